// gfx950 front-end kernels: padded pyramid build, per-track pyramidal Lucas–Kanade, fused corner detectors.
// Replaces the OpenCV calls behind BaseFeatureMatcher / BaseFeatureExtractor (see include/pmv_hip.h for file:line).
// All integer results (track coordinates after truncation, status bytes, corner lists) are bit-exact with oracle/:
//   * LK: every window sum is an exact integer (per-lane int32 partials, 64-bit wave reduction), rounded once to f32;
//   * GFTT: float32 ops in a fixed order, compiled with -ffp-contract=off; the 3x3 box sum is a 9-term double sum.
#include "pmv_device.h"
#include "pmv_prof.h"
#include <float.h>

namespace pmv {

// =========================================================================================================
// Pyramid
// =========================================================================================================
// Both kernels work on ONE padded output row per workgroup (grid = (1, padded rows, frames)) and touch global memory only with
// coalesced dword / 16-byte accesses; the byte-granular REFLECT_101 gather happens in LDS. (The first version read 25 bytes per
// output pixel straight from global memory: 0.6 TB/s; this one moves each source row once per output row that needs it.)

// NB valid bytes starting at byte `off` of a 4-byte-aligned LDS row as packed little-endian dwords (see k_lk)
template <int NB>
__device__ inline void lds_bytes_aligned(const uint8_t* row, int off, uint32_t (&e)[(NB + 3) / 4]) {
    constexpr int ND = (NB + 3 + 3) / 4, NE = (NB + 3) / 4;
    const uint32_t* w = (const uint32_t*)row + (off >> 2);
    const uint32_t sh = (uint32_t)off & 3u;
    uint32_t d[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) d[j] = w[j];
#pragma unroll
    for (int j = 0; j < NE; j++) e[j] = __builtin_amdgcn_alignbyte(j + 1 < ND ? d[j + 1] : 0u, d[j], sh);
}

// One padded REFLECT_101 row from an unpadded row of `w` bytes that sits in LDS at byte `lead` of a 4-byte-aligned buffer (with >= 24
// bytes of slack behind it): 16 bytes per thread and step. Interior groups are aligned gathers; the left and right borders are
// byte-reversed runs of the row (PAD % 16 == 0, so only the group that straddles the right edge mixes both). Rows narrower than
// PAD + 48 take the per-byte path everywhere.
template <int T>
__device__ inline void write_padded_row(const uint8_t* lrow, int lead, int w, uint8_t* dst) {
    const int pw = w + 2 * PAD;
    const bool wide = w >= PAD + 48;   // block-uniform
    for (int x16 = threadIdx.x * 16; x16 < pw; x16 += T * 16) {   // (strides are multiples of 64: a 16-byte store never leaves the row)
        uint32_t v[4];
        const int sx0 = x16 - PAD;
        if (sx0 >= 0 && sx0 + 15 < w) {
            lds_bytes_aligned<16>(lrow, lead + sx0, v);
        } else if (wide && (sx0 < 0 || sx0 >= w)) {
            // dst[x16 + k] = row[s - k]: left border s = PAD - x16 (>= 16), right border s = 2w - 2 - sx0
            const int s = sx0 < 0 ? -sx0 : 2 * w - 2 - sx0;
            uint32_t e[4];
            lds_bytes_aligned<16>(lrow, lead + s - 15, e);
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = __builtin_bswap32(e[3 - q]);
        } else if (wide) {
            // the group that straddles the right edge: the first n_in bytes run forward, the rest is the reversed run
            const int n_in = w - sx0;   // 1 .. 15
            uint32_t a[4], e[4];
            lds_bytes_aligned<16>(lrow, lead + sx0, a);
            lds_bytes_aligned<16>(lrow, lead + 2 * w - 2 - sx0 - 15, e);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lo = n_in - 4 * q;
                const uint32_t m = lo >= 4 ? 0xffffffffu : (lo <= 0 ? 0u : ((1u << (8 * lo)) - 1u));
                v[q] = (a[q] & m) | (__builtin_bswap32(e[3 - q]) & ~m);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t t = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) t |= (uint32_t)lrow[lead + reflect101(sx0 + 4 * q + k, w)] << (8 * k);
                v[q] = t;
            }
        }
        *(uint4*)(dst + x16) = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// level 0: the gray frame sits in the interior of its padded buffer (staged there row by row); this kernel adds the REFLECT_101 frame
// IN PLACE: every padded row is rebuilt from the interior row it mirrors (an interior row from itself: same bytes; a border row from
// another row's interior, which no block changes).
constexpr int PAD0_T = 128, PAD0_R = 4;   // rows per workgroup: their loads are in flight together
__host__ __device__ inline int pad0_row_lds(int w) { return ((w + 3 + 24 + 3) / 4) * 4; }
// `tight` != nullptr: the frames come from a buffer of tight gray frames (frame z at tight + z * w * h: staging / upload / ingest; rows
// generally unaligned, fetched as the aligned dwords that cover them) instead of from the level's own interior.
__global__ __launch_bounds__(PAD0_T) void k_pad_level0(uint8_t* slots, PyrLayout L, int first_slot, const uint8_t* __restrict__ tight) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srow_all[];   // PAD0_R x row_lds bytes
    uint8_t* slot = slots + (size_t)(first_slot + blockIdx.z) * L.slot_bytes;
    const int w = L.w[0], h = L.h[0], stride = L.stride[0];
    const int row_lds = pad0_row_lds(w);
    const int ph = h + 2 * PAD;
    const int py0 = blockIdx.y * PAD0_R;
    unsigned lead[PAD0_R];
#pragma unroll
    for (int r = 0; r < PAD0_R; r++) {
        const int py = py0 + r < ph ? py0 + r : ph - 1;     // (rows past the end repeat the last one; they are not stored)
        const int sy = reflect101(py - PAD, h);
        const uint32_t* src;
        int nd;
        if (tight) {
            const size_t b = (size_t)blockIdx.z * (size_t)w * (size_t)h + (size_t)sy * (size_t)w;   // byte offset of the source row (the buffer is 256-byte aligned and has slack behind it)
            lead[r] = (unsigned)(b & 3u);
            src = (const uint32_t*)(tight + (b & ~(size_t)3));
            nd = (int)((lead[r] + (unsigned)w + 3u) >> 2);
        } else {
            lead[r] = 0u;                                   // interior rows start 64-byte aligned (PAD and the stride are multiples of 64)
            src = (const uint32_t*)(slot + L.gray_off + (size_t)sy * (size_t)stride);
            nd = (w + 3) >> 2;                              // (the last dword may reach into the right border: inside the buffer)
        }
        uint32_t* srow = (uint32_t*)(srow_all + r * row_lds);
        for (int d = threadIdx.x; d < nd; d += PAD0_T) srow[d] = src[d];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PAD0_R; r++) {
        if (py0 + r >= ph) break;
        write_padded_row<PAD0_T>(srow_all + r * row_lds, (int)lead[r], w, slot + L.off[0] + (size_t)(py0 + r) * stride);
    }
}

// level l-1 (padded) -> level l (padded): cv::pyrDown [1 4 6 4 1]^2, (sum+128)>>8. A workgroup produces PYR_R padded output rows: per
// row the vertical taps straight from coalesced dword loads of the five source rows (source columns -4 .. 2*dw+3: the level origin
// is 4-byte aligned) as packed 16-bit sums into LDS, then the horizontal taps of the dw INTERIOR pixels in packed 16-bit arithmetic
// (every intermediate <= 16 * 4080 + 128 < 2^16), then the padded row as mirrored runs of that interior row (the pyrDown value at a
// REFLECT_101-mapped destination coordinate IS the interior pixel at that coordinate). Integer sums: the order does not matter.
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
__device__ inline pk_u16 as_pk(uint32_t x) { union { uint32_t u; pk_u16 p; } c; c.u = x; return c.p; }
__device__ inline uint32_t as_u32(pk_u16 x) { union { uint32_t u; pk_u16 p; } c; c.p = x; return c.u; }
constexpr int PYR_T = 256, PYR_R = 4;   // output rows per workgroup
__host__ __device__ inline int pyr_vs_row(int dw) { return (2 * dw + 24 + 3) & ~3; }        // uint16 entries
__host__ __device__ inline int pyr_out_row(int dw) { return ((dw + 3 + 24 + 3) / 4) * 4; }  // bytes
__global__ __launch_bounds__(PYR_T) void k_pyrdown(uint8_t* slots, PyrLayout L, int ld, int first_slot) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pyr_lds[];
    uint8_t* slot = slots + (size_t)(first_slot + blockIdx.z) * L.slot_bytes;
    const int ls = ld - 1;
    const uint8_t* src = level_origin((const uint8_t*)slot, L, ls);
    const int ss = L.stride[ls];
    const int dw = L.w[ld], dh = L.h[ld], ds = L.stride[ld];
    const int ph = dh + 2 * PAD;
    const int py0 = blockIdx.y * PYR_R;
    const int nd = (2 * dw + 8) >> 2;
    const int vs_row = pyr_vs_row(dw), out_row = pyr_out_row(dw);
    uint8_t* orow_all = pyr_lds + (size_t)PYR_R * vs_row * 2;
#pragma unroll
    for (int r = 0; r < PYR_R; r++) {
        const int py = py0 + r < ph ? py0 + r : ph - 1;
        const int ry = reflect101(py - PAD, dh);
        const uint8_t* r0 = src + (ptrdiff_t)(2 * ry - 2) * ss - 4;
        uint16_t* vs = (uint16_t*)pyr_lds + r * vs_row;      // vs[c + 4] = vertical sum of source column c
        for (int d = threadIdx.x; d < nd; d += PYR_T) {
            const uint32_t a = *(const uint32_t*)(r0 + 4 * d), b = *(const uint32_t*)(r0 + ss + 4 * d), c = *(const uint32_t*)(r0 + 2 * ss + 4 * d),
                           e = *(const uint32_t*)(r0 + 3 * ss + 4 * d), f = *(const uint32_t*)(r0 + 4 * ss + 4 * d);
            // bytes 0,2 and bytes 1,3 of each row as 16-bit pairs; (a + f) + 4 (b + e) + 6 c <= 4080
            const uint32_t M = 0x00ff00ffu;
            const pk_u16 lo = (as_pk(a & M) + as_pk(f & M)) + (pk_u16)(4) * (as_pk(b & M) + as_pk(e & M)) + (pk_u16)(6) * as_pk(c & M);
            const pk_u16 hi = (as_pk((a >> 8) & M) + as_pk((f >> 8) & M)) + (pk_u16)(4) * (as_pk((b >> 8) & M) + as_pk((e >> 8) & M)) + (pk_u16)(6) * as_pk((c >> 8) & M);
            const uint32_t l = as_u32(lo), hh = as_u32(hi);   // l = (col0, col2), hh = (col1, col3)
            *(uint2*)(vs + 4 * d) = make_uint2((l & 0xffffu) | (hh << 16), (l >> 16) | (hh & 0xffff0000u));
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PYR_R; r++) {
        // D[j] = (E[j], O[j]) = vertical sums of source columns 2j, 2j+1 = dword j + 2 of the row; output pixel i:
        // E[i-1] + E[i+1] + 4 (O[i-1] + O[i]) + 6 E[i]
        const uint32_t* vsd = (const uint32_t*)((const uint16_t*)pyr_lds + r * vs_row) + 2;
        uint32_t* orow = (uint32_t*)(orow_all + r * out_row);
        for (int i = threadIdx.x * 4; i < dw; i += PYR_T * 4) {
            uint32_t D[7];
#pragma unroll
            for (int q = 0; q < 7; q++) D[q] = vsd[i - 1 + q];   // D[q] = pair i - 1 + q
            uint32_t res[2];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int o = 2 * half;   // outputs i + o, i + o + 1
                const pk_u16 Em = as_pk(__builtin_amdgcn_perm(D[o + 1], D[o], 0x05040100u)), E0 = as_pk(__builtin_amdgcn_perm(D[o + 2], D[o + 1], 0x05040100u)),
                             Ep = as_pk(__builtin_amdgcn_perm(D[o + 3], D[o + 2], 0x05040100u));
                const pk_u16 Om = as_pk(__builtin_amdgcn_perm(D[o + 1], D[o], 0x07060302u)), O0 = as_pk(__builtin_amdgcn_perm(D[o + 2], D[o + 1], 0x07060302u));
                const pk_u16 t = ((Em + Ep) + (pk_u16)(4) * (Om + O0) + (pk_u16)(6) * E0 + (pk_u16)(128)) >> (pk_u16)(8);
                res[half] = as_u32(t);   // two bytes in the 16-bit lanes
            }
            orow[i >> 2] = (res[0] & 0xffu) | ((res[0] >> 8) & 0xff00u) | ((res[1] & 0xffu) << 16) | ((res[1] & 0xff0000u) << 8);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PYR_R; r++) {
        if (py0 + r >= ph) break;
        write_padded_row<PYR_T>(orow_all + r * out_row, 0, dw, slot + L.off[ld] + (size_t)(py0 + r) * ds);
    }
}

// cv::cvtColor(BGR2GRAY) for 8-bit images (Frame::init, Frame.cpp:40-41): gray = (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14, the 14-bit
// fixed-point form of 0.114 B + 0.587 G + 0.299 R [mem: OpenCV 3.4 color.cpp]; identity for B = G = R (KITTI's gray PNGs, quirk Q2).
// One thread per 4 output pixels: 12 source bytes as three dword loads when the row is 4-byte aligned, byte loads otherwise.
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* __restrict__ bgr, int w, int h, int stride, uint8_t* __restrict__ gray) {
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= w || y >= h) return;
    const uint8_t* row = bgr + (size_t)y * stride + 3 * (size_t)x4;
    uint8_t out[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (x4 + k >= w) { out[k] = 0; continue; }
        const int b = row[3 * k], g = row[3 * k + 1], r = row[3 * k + 2];
        out[k] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
    uint8_t* dst = gray + (size_t)y * w + x4;
#pragma unroll
    for (int k = 0; k < 4; k++) if (x4 + k < w) dst[k] = out[k];
}
hipError_t launch_bgr2gray(hipStream_t s, const uint8_t* d_bgr, int w, int h, int stride, uint8_t* d_gray) {
    if (!d_bgr || !d_gray || w < 1 || h < 1 || stride < 3 * w) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bgr2gray, dim3((w + 1023) / 1024, h), dim3(256), 0, s, d_bgr, w, h, stride, d_gray);
    return hipGetLastError();
}

hipError_t launch_pad_level0(hipStream_t s, uint8_t* slots, const PyrLayout& L, int first_slot, int n, const uint8_t* tight) {
    if (!slots || n < 1 || first_slot < 0 || L.n_levels < 1) return hipErrorInvalidValue;
    dim3 grid(1, (L.h[0] + 2 * PAD + PAD0_R - 1) / PAD0_R, n);
    const size_t shm = (size_t)pad0_row_lds(L.w[0]) * PAD0_R;
    ProfScope ps(K_PAD0, s);
    hipLaunchKernelGGL(k_pad_level0, grid, dim3(PAD0_T), shm, s, slots, L, first_slot, tight);
    return hipGetLastError();
}
hipError_t launch_pyrdown(hipStream_t s, uint8_t* slots, const PyrLayout& L, int ld, int first_slot, int n) {
    if (!slots || n < 1 || first_slot < 0 || ld < 1 || ld >= L.n_levels) return hipErrorInvalidValue;
    dim3 grid(1, (L.h[ld] + 2 * PAD + PYR_R - 1) / PYR_R, n);
    const size_t shm = ((size_t)pyr_vs_row(L.w[ld]) * 2 + (size_t)pyr_out_row(L.w[ld])) * PYR_R;
    ProfScope ps(K_PYRDOWN, s);
    hipLaunchKernelGGL(k_pyrdown, grid, dim3(PYR_T), shm, s, slots, L, ld, first_slot);
    return hipGetLastError();
}

// =========================================================================================================
// Pyramidal Lucas–Kanade: one 256-thread block (4 wavefronts) per track, all levels inside one launch.
//   thread = window row (tid >> 3), 4 consecutive columns; its (I, Ix, Iy) samples live in registers for the level.
//   LDS per block: 35x35 u8 tile of the previous image (-> Scharr on the fly, no derivative image in HBM),
//   33x33 (dx,dy) int16 pairs, and a 64x64 u8 search tile of the next image that is re-staged only when the
//   window walks out of it.
// =========================================================================================================
constexpr int SI_STRIDE = 40;   // 40 staged bytes per row: 35 tile columns + up to 3 bytes of dword-alignment slack
constexpr int SD_STRIDE = 33;
constexpr int SJ_STRIDE = 68;

__device__ inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ inline void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
    w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    w10 = __float2int_rn((1.f - a) * b * 16384.f);
    w11 = 16384 - w00 - w01 - w10;
}

// Block-wide EXACT sums of per-thread int32 partials (4 wavefronts). A wavefront's total leaves wave_sum_i32_exact as a double: every
// sum of these integers stays far below 2^53 (<= 1024 pixels x 8160 x 4080 = 3.4e10), so double addition is exact in any order.
// Each wavefront leaves its total in LDS and after ONE barrier every thread adds the four totals. `slot` alternates between calls that are not separated by another barrier (iteration parity),
// so the next write never overtakes a pending read. (float)(total) rounds the exact integer once, like (float)(int64).
// T = threads per track: 256 (four wavefronts: the shortest chain per track, used when one sequence's ~300 tracks are all there is)
// or 64 (one wavefront: no barrier in the reduction and four times as many tracks resident per CU — the throughput form used by the
// batched launch, where thousands of tracks are in flight). Integer sums are exact, so both give the same bits.
constexpr int LK_T = 256;
// One wavefront's exact sum of int32 partials as a double: the partial is split into its low 16 bits (unsigned) and the rest (signed),
// each half summed with six DPP steps folded into v_add_u32 and read from lane 63 (no half can overflow: 64 x 65535 and
// 64 x 2^15), total = 65536 * H + L, exact in double. The double-precision DPP form this replaces cost 23 dependent vector instructions
// per value (1.4 k clk per LK iteration for two values: a fifth of a track's time in the batched launch).
// The halves of all N values go through each step TOGETHER: 2N independent instructions per step fill the wait states a DPP or swap
// result needs before its next use (one chain after the other spent more slots on s_nop than on adds).
template <int N>
__device__ __forceinline__ void wave_sum_i32_exact(const int (&part)[N], double (&out)[N]) {
    int h[2 * N];
#pragma unroll
    for (int k = 0; k < N; k++) { h[2 * k] = part[k] & 0xffff; h[2 * k + 1] = part[k] >> 16; }
#pragma unroll
    for (int k = 0; k < 2 * N; k++) h[k] += __builtin_amdgcn_update_dpp(0, h[k], 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < 2 * N; k++) h[k] += __builtin_amdgcn_update_dpp(0, h[k], 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < 2 * N; k++) h[k] += __builtin_amdgcn_update_dpp(0, h[k], 0x141, 0xf, 0xf, false);   // row_half_mirror
#pragma unroll
    for (int k = 0; k < 2 * N; k++) h[k] += __builtin_amdgcn_update_dpp(0, h[k], 0x140, 0xf, 0xf, false);   // row_mirror: the 16-lane row's sum in every lane
    // across the four rows: lane 15 of a row into the next row (rows 1 and 3), then lane 31 into rows 2 and 3 - lane 63 holds the total
#pragma unroll
    for (int k = 0; k < 2 * N; k++) h[k] += __builtin_amdgcn_update_dpp(0, h[k], 0x142, 0xa, 0xf, false);   // row_bcast:15
#pragma unroll
    for (int k = 0; k < 2 * N; k++) h[k] += __builtin_amdgcn_update_dpp(0, h[k], 0x143, 0xc, 0xf, false);   // row_bcast:31
#pragma unroll
    for (int k = 0; k < N; k++) out[k] = (double)__builtin_amdgcn_readlane(h[2 * k + 1], 63) * 65536.0 + (double)__builtin_amdgcn_readlane(h[2 * k], 63);
}
template <int N, int T>
__device__ inline void block_sum_exact(const int (&part)[N], double (&out)[N], double* sred /* [2][T/64][4] */, int slot) {
    constexpr int NW = T / 64;
    if (NW == 1) { wave_sum_i32_exact<N>(part, out); return; }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v[N];
    wave_sum_i32_exact<N>(part, v);
#pragma unroll
    for (int k = 0; k < N; k++)
        if (lane == 0) sred[(slot * NW + wv) * 4 + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        double a = sred[(slot * NW + 0) * 4 + k];
#pragma unroll
        for (int w = 1; w < NW; w++) a += sred[(slot * NW + w) * 4 + k];
        out[k] = a;
    }
}

// NB consecutive bytes that start at byte `off` (any alignment) of an LDS row whose base is 4-byte aligned, fetched as aligned dword
// loads and shifted into place: e[j] holds bytes 4j .. 4j+3 of the run. A byte load serves one pixel per LDS instruction; this serves
// four, and with lanes = distinct rows of an odd dword stride the dword loads are free of bank conflicts. `off & 3` must be
// wave-uniform in practice (it is: tile origin + a multiple of 4), but nothing here depends on it.
template <int NB>
__device__ inline void lds_row_bytes(const uint8_t* row, int off, uint32_t (&e)[(NB + 3) / 4]) {
    constexpr int ND = (NB + 3 + 3) / 4, NE = (NB + 3) / 4;
    const uint32_t* w = (const uint32_t*)row + (off >> 2);
    const uint32_t sh = (uint32_t)off & 3u;
    uint32_t d[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) d[j] = w[j];
#pragma unroll
    for (int j = 0; j < NE; j++) e[j] = __builtin_amdgcn_alignbyte(j + 1 < ND ? d[j + 1] : 0u, d[j], sh);
}
#define PACKED_BYTE(e, k) ((int)(((e)[(k) >> 2] >> (((k) & 3) * 8)) & 0xffu))

// Bilinear sample as two packed dot products (v_dot2_i32_i16): the pixel of the upper row and the pixel of the lower row of one column
// share a register as two 16-bit lanes, the weights of that column likewise: w00*p0 + w10*p1 is ONE instruction, the neighbouring
// column adds w01*p0n + w11*p1n. Same integer as the four 24-bit products (|terms| < 2^28).
typedef short pk_s16 __attribute__((ext_vector_type(2)));
__device__ inline pk_s16 as_s16x2(uint32_t x) { union { uint32_t u; pk_s16 p; } c; c.u = x; return c.p; }
__device__ inline uint32_t pack_weights(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
// column k of two packed byte rows -> (top | bottom << 16)
#define PACKED_COLUMN(top, bot, k) __builtin_amdgcn_perm((bot)[(k) >> 2], (top)[(k) >> 2], 0x0c040c00u | ((uint32_t)((k) & 3) << 16) | (uint32_t)((k) & 3))
__device__ inline int dot2_acc(uint32_t a, uint32_t w, int acc) { return __builtin_amdgcn_sdot2(as_s16x2(a), as_s16x2(w), acc, false); }

// 64x64-byte search tile (stride % 64 == 0, PAD % 4 == 0, tx0 % 4 == 0) in two halves: `tile_issue` starts the global loads (16 bytes
// per lane and load, 256 / T of them) into registers, `tile_commit` writes them to LDS. The caller puts independent work between the
// two - the iterations of the level above - so that the round trip to HBM (7 us per tile in a batched launch, a third of a track's
// time when it was waited for in place) overlaps it.
struct __attribute__((packed, aligned(4))) U4A { uint32_t x, y, z, w; };   // 16 bytes at a 4-byte-aligned address: one global_load_dwordx4
struct __attribute__((packed, aligned(4))) U2A { uint32_t x, y; };
template <int T>
__device__ __forceinline__ void tile_issue(U4A (&reg)[256 / T], const uint8_t* Jorg, int js, int tx0, int ty0, int tid) {
#pragma unroll
    for (int q = 0; q < 256 / T; q++) {
        const int idx = tid + q * T, row = idx >> 2, x4 = idx & 3;
        reg[q] = *(const U4A*)(Jorg + (ptrdiff_t)(ty0 + row) * js + tx0 + 16 * x4);
    }
}
template <int T>
__device__ __forceinline__ void tile_commit(uint8_t* sJ, const U4A (&reg)[256 / T], int tid) {
#pragma unroll
    for (int q = 0; q < 256 / T; q++) {
        const int idx = tid + q * T, row = idx >> 2, x4 = idx & 3;
        uint32_t* d = (uint32_t*)(sJ + row * SJ_STRIDE) + 4 * x4;
        d[0] = reg[q].x; d[1] = reg[q].y; d[2] = reg[q].z; d[3] = reg[q].w;
    }
}
template <int T>
__device__ inline void stage_J(uint8_t* sJ, const uint8_t* Jorg, int js, int tx0, int ty0, int tid) {
    U4A reg[256 / T];
    tile_issue<T>(reg, Jorg, js, tx0, ty0, tid);
    tile_commit<T>(sJ, reg, tid);
}

// One 256-thread block (4 wavefronts) per track: thread = window row (tid >> 3) and 4 consecutive columns ((tid & 7) * 4); its
// (I, Ix, Iy) samples live in registers for the whole level. A single wavefront per track is VALU-issue-bound (2.2 k cycles per
// iteration for 16 pixels per lane); four wavefronts on the four SIMDs of a CU cut the per-iteration chain to 4 pixels per lane
// plus one barrier. All sums are integers (order-free), so the result is bit-identical to the sequential algorithm.
// One track through all pyramid levels; executed by one whole 256-thread block (every thread gets the same results). `stamp_on`:
// this block feeds the diagnostic phase timers.
struct LKResult { float x, y, err; int status, n_iter, n_lev; };
template <int T, bool STAMPS>
__device__ __forceinline__ LKResult lk_track_block(const uint8_t* __restrict__ prevS, const uint8_t* __restrict__ nextS, const PyrLayout& L,
                                                   const float px0, const float py0, const LKParams& P, const bool stamp_on) {
    constexpr int NPRE = 4;                       // I tiles staged per group (all levels of a 4-level pyramid at once)
    constexpr int SI_BYTES = 35 * SI_STRIDE + 8;   // + two dwords: the aligned loads of the last row may touch the first
    __shared__ __attribute__((aligned(16))) uint8_t sIall[NPRE * SI_BYTES];
    __shared__ __attribute__((aligned(16))) short2 sD[33 * SD_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t sJ[64 * SJ_STRIDE];
    __shared__ double sred[2 * (T / 64) * 4];
    constexpr int TPR = T / 32, PP = 32 / TPR;   // threads per window row, pixels per thread (256 -> 8 x 4, 64 -> 2 x 16)
    constexpr int NIQ = (35 * 5 + T - 1) / T;     // 8-byte loads per thread of one I tile (35 rows x 40 bytes)
    const int tid = threadIdx.x;
    // 32 consecutive lanes = the 32 window rows of one column group: their LDS rows differ, and with the odd dword strides of sJ (17)
    // and sD (33) they fall into 32 different banks (the sums are exact integers, so the pixel-to-lane assignment is free)
    const int r = tid & 31, c0 = (tid >> 5) * PP;
    const int W = LK_WIN;
    const float half = 15.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    float outx = 0.f, outy = 0.f, err = 0.f;
    int status = 1;
    int slot = 0;
    int n_iter = 0, n_lev = 0;   // work counters for the roofline's measured OPS_lk
    const int ml = L.n_levels - 1;
    unsigned long long t_prev = __builtin_readcyclecounter();
    const unsigned long long wall0 = (STAMPS && P.stamps && stamp_on) ? wall_clock64() : 0ull;   // 100 MHz: calibrates the cycle counter of the stamps
#define LSTAMP(k) do { if (STAMPS && P.stamps && stamp_on && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&P.stamps[k], t_ - t_prev); t_prev = t_; } } while (0)

    // where the template window of a level sits: known for every level before anything is tracked (only the search side depends on the
    // level above), so the I tiles of all levels are fetched at once - one round trip to memory instead of one per level
    auto template_origin = [&](int level, float& fx, float& fy, int& ipx, int& ipy) -> bool {
        const float lscale = __int_as_float((127 - level) << 23);   // 2^-level, the value of (float)(1. / (1 << level))
        fx = px0 * lscale - half; fy = py0 * lscale - half;
        ipx = (int)floorf(fx); ipy = (int)floorf(fy);
        return !(ipx < -W || ipx >= L.w[level] || ipy < -W || ipy >= L.h[level]);
    };
    // 35x35 I tiles (rows ipy-1 .. ipy+33, cols ipx-1 .. ipx+33, always inside the padded buffer) of levels top .. top-3 as 8-byte copies
    // of 40-byte rows that start at the 4-byte-aligned column below ipx-1 (35 + 3 <= 40 columns stay inside the 64-pixel frame)
    int tx0 = 0, ty0 = 0, tile_level = -1;   // the search tile in sJ: origin and the level it was cut from
    auto stage_I_group = [&](int top, bool with_search_tile) {
        U2A ireg[NPRE][NIQ];
        U4A jreg[256 / T];
        bool j_ok = false;
#pragma unroll
        for (int g = 0; g < NPRE; g++) {
            const int lv = top - g;
            float fx, fy; int ipx, ipy;
            if (lv < 0 || !template_origin(lv, fx, fy, ipx, ipy)) continue;   // block-uniform
            const uint8_t* Iorg = level_origin(prevS, L, lv);
            const int ls = L.stride[lv], ax0 = (ipx - 1) & ~3;
#pragma unroll
            for (int q = 0; q < NIQ; q++) {
                const int idx = tid + q * T, y = idx / 5, xw = idx - y * 5;
                if (idx < 35 * 5) ireg[g][q] = *(const U2A*)(Iorg + (ptrdiff_t)(ipy - 1 + y) * ls + ax0 + 8 * xw);
            }
        }
        if (with_search_tile) {   // the top level's search starts at the template position: its tile rides along
            float fx, fy; int inx, iny;
            if (template_origin(top, fx, fy, inx, iny)) {
                tx0 = (inx - 16) & ~3; ty0 = iny - 16;
                tile_issue<T>(jreg, level_origin(nextS, L, top), L.stride[top], tx0, ty0, tid);
                j_ok = true;
            }
        }
#pragma unroll
        for (int g = 0; g < NPRE; g++) {
            const int lv = top - g;
            float fx, fy; int ipx, ipy;
            if (lv < 0 || !template_origin(lv, fx, fy, ipx, ipy)) continue;
            uint8_t* dst = sIall + (lv & (NPRE - 1)) * SI_BYTES;
#pragma unroll
            for (int q = 0; q < NIQ; q++) {
                const int idx = tid + q * T, y = idx / 5, xw = idx - y * 5;
                if (idx < 35 * 5) { uint32_t* d = (uint32_t*)(dst + y * SI_STRIDE) + 2 * xw; d[0] = ireg[g][q].x; d[1] = ireg[g][q].y; }
            }
        }
        if (j_ok) { tile_commit<T>(sJ, jreg, tid); tile_level = top; }
    };
    stage_I_group(ml, true);
    LSTAMP(1);

    for (int level = ml; level >= 0; level--) {
        LSTAMP(0);
        const int lw = L.w[level], lh = L.h[level], ls = L.stride[level];
        const uint8_t* Jorg = level_origin(nextS, L, level);
        __syncthreads();
        if (level != ml && ((ml - level) & (NPRE - 1)) == 0) {   // pyramids deeper than NPRE levels: the next group of template tiles
            stage_I_group(level, false);
            __syncthreads();
            LSTAMP(1);
        }
        const uint8_t* sI = sIall + (level & (NPRE - 1)) * SI_BYTES;
        float prevx, prevy;
        int ipx, ipy;
        const bool inside = template_origin(level, prevx, prevy, ipx, ipy);
        float nx, ny;
        if (level == ml) { const float lscale = __int_as_float((127 - level) << 23); nx = px0 * lscale; ny = py0 * lscale; }
        else { nx = outx * 2.f; ny = outy * 2.f; }
        outx = nx; outy = ny;
        if (!inside) {   // block-uniform
            if (level == 0) { status = 0; err = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        bilinear_weights(prevx - ipx, prevy - ipy, iw00, iw01, iw10, iw11);
        const int aoff = (ipx - 1) - ((ipx - 1) & ~3);
        // (tile origin (ipx-1, ipy-1) = byte aoff of row 0)
        LSTAMP(1);
        // ---- Scharr (calcSharrDeriv) at the 33x33 sample positions; constant 0 outside the image. Task = (row y, run of SEG columns):
        // three rows of SEG + 2 tile bytes through aligned dword loads, then the 3x3 stencil slides along the run in registers.
        {
            constexpr int SEG = (T >= 256) ? 5 : 11, NSEG = (33 + SEG - 1) / SEG;
            for (int idx = tid; idx < 33 * NSEG; idx += T) {
                const int sg = idx / 33, y = idx - sg * 33, x0 = sg * SEG;
                uint32_t ra[(SEG + 2 + 3) / 4], rb[(SEG + 2 + 3) / 4], rc[(SEG + 2 + 3) / 4];
                lds_row_bytes<SEG + 2>(sI + y * SI_STRIDE, aoff + x0, ra);
                lds_row_bytes<SEG + 2>(sI + (y + 1) * SI_STRIDE, aoff + x0, rb);
                lds_row_bytes<SEG + 2>(sI + (y + 2) * SI_STRIDE, aoff + x0, rc);
                const int gy = ipy + y;
                const bool row_in = gy >= 0 && gy < lh;
                // per tile column: the vertical smoothing 3*(a + c) + 10*b (for d/dx) and the vertical difference c - a (for d/dy) once,
                // then dx = S[x+2] - S[x], dy = 3*(V[x] + V[x+2]) + 10*V[x+1]: the same integers as the 3x3 stencil per pixel
                int S[SEG + 2], V[SEG + 2];
#pragma unroll
                for (int k = 0; k < SEG + 2; k++) {
                    const int a = PACKED_BYTE(ra, k), b = PACKED_BYTE(rb, k), c = PACKED_BYTE(rc, k);
                    S[k] = (a + c) * 3 + b * 10;
                    V[k] = c - a;
                }
#pragma unroll
                for (int k = 0; k < SEG; k++) {
                    const int x = x0 + k;
                    if (x < 33) {
                        const int gx = ipx + x;
                        short2 d = make_short2(0, 0);
                        if (row_in && gx >= 0 && gx < lw) {
                            d.x = (short)(S[k + 2] - S[k]);
                            d.y = (short)((V[k + 2] + V[k]) * 3 + V[k + 1] * 10);
                        }
                        sD[y * SD_STRIDE + x] = d;
                    }
                }
            }
        }
        __syncthreads();
        LSTAMP(2);
        // ---- this thread's PP window samples (I with 5 fractional bits, Ix, Iy) + exact A sums
        // Ix / Iy of two neighbouring pixels share a register as 16-bit halves (|derivative sample| <= 4080): the window sums then take one
        // v_dot2_i32_i16 per pixel PAIR instead of a 24-bit multiply and an add per pixel - the same exact integers
        int Iv[PP];
        uint32_t IxP[PP / 2], IyP[PP / 2];
        int apart[3] = {0, 0, 0};
        {
            uint32_t ia[(PP + 1 + 3) / 4], ib[(PP + 1 + 3) / 4];
            lds_row_bytes<PP + 1>(sI + (r + 1) * SI_STRIDE, aoff + c0 + 1, ia);
            lds_row_bytes<PP + 1>(sI + (r + 2) * SI_STRIDE, aoff + c0 + 1, ib);
            const uint32_t* d0 = (const uint32_t*)&sD[r * SD_STRIDE + c0];   // (dx | dy << 16) per pixel
            const uint32_t W0 = pack_weights(iw00, iw10), W1 = pack_weights(iw01, iw11);   // (upper, lower) weights of a column / of its right neighbour
            uint32_t pc = PACKED_COLUMN(ia, ib, 0);
            uint32_t q0 = d0[0], q1 = d0[SD_STRIDE];
            uint32_t xc = __builtin_amdgcn_perm(q1, q0, 0x05040100u), yc = __builtin_amdgcn_perm(q1, q0, 0x07060302u);   // (upper.dx, lower.dx), (upper.dy, lower.dy)
#pragma unroll
            for (int k = 0; k < PP; k++) {
                const uint32_t pn = PACKED_COLUMN(ia, ib, k + 1);
                const uint32_t q0n = d0[k + 1], q1n = d0[SD_STRIDE + k + 1];
                const uint32_t xn = __builtin_amdgcn_perm(q1n, q0n, 0x05040100u), yn = __builtin_amdgcn_perm(q1n, q0n, 0x07060302u);
                // kept as 256 - 512*I: the accumulator start of the search-side sample, so that (J_sum + 256 - 512*I) >> 9 = descale(J_sum, 9) - I
                // needs neither the rounding constant nor the subtraction per pixel and iteration (512*I is a multiple of 2^9: exact)
                Iv[k] = (1 << 8) - ((dot2_acc(pn, W1, dot2_acc(pc, W0, 1 << 8)) >> 9) << 9);
                const int ixv = dot2_acc(xn, W1, dot2_acc(xc, W0, 1 << 13)) >> 14;
                const int iyv = dot2_acc(yn, W1, dot2_acc(yc, W0, 1 << 13)) >> 14;
                if (k & 1) {
                    IxP[k >> 1] = __builtin_amdgcn_perm((uint32_t)ixv, IxP[k >> 1], 0x05040100u);   // (even pixel | odd pixel << 16)
                    IyP[k >> 1] = __builtin_amdgcn_perm((uint32_t)iyv, IyP[k >> 1], 0x05040100u);
                    apart[0] = dot2_acc(IxP[k >> 1], IxP[k >> 1], apart[0]);
                    apart[1] = dot2_acc(IxP[k >> 1], IyP[k >> 1], apart[1]);
                    apart[2] = dot2_acc(IyP[k >> 1], IyP[k >> 1], apart[2]);
                } else { IxP[k >> 1] = (uint32_t)ixv; IyP[k >> 1] = (uint32_t)iyv; }
                pc = pn; xc = xn; yc = yn;
            }
        }
        double sA[3];
        block_sum_exact<3, T>(apart, sA, sred, slot); slot ^= 1;
        const float A11 = (float)sA[0] * FLT_SCALE, A12 = (float)sA[1] * FLT_SCALE, A22 = (float)sA[2] * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * W * W);
        if (minEig < P.min_eig || D < FLT_EPSILON) {   // block-uniform (computed from block-wide sums)
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        LSTAMP(3);
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        n_lev++;
        // Speculative fetch of the NEXT level's search tile around twice this level's start: the loads fly while this level iterates
        // and are written to LDS after the loop. The iterations of a level move the window by a pixel or two, the 64-pixel tile has
        // 15 pixels of slack on every side; when the guess misses, the loop below re-stages as before - the result never depends on it.
        U4A jreg[256 / T];
        bool pre = false;
        int ptx0 = 0, pty0 = 0;
        if (level > 0) {
            const float gx = (nx + half) * 2.f - half, gy = (ny + half) * 2.f - half;
            const int pinx = (int)floorf(gx), piny = (int)floorf(gy);
            if (!(pinx < -W || pinx >= L.w[level - 1] || piny < -W || piny >= L.h[level - 1])) {
                ptx0 = (pinx - 16) & ~3; pty0 = piny - 16;
                tile_issue<T>(jreg, level_origin(nextS, L, level - 1), L.stride[level - 1], ptx0, pty0, tid);
                pre = true;
            }
        }

        for (int j = 0; j < P.max_iter; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -W || inx >= lw || iny < -W || iny >= lh) {
                if (level == 0) status = 0;
                break;
            }
            int wx = inx - tx0, wy = iny - ty0;
            if (tile_level != level || wx < 0 || wx > 31 || wy < 0 || wy > 31) {
                tx0 = (inx - 16) & ~3; ty0 = iny - 16;
                __syncthreads();
                stage_J<T>(sJ, Jorg, ls, tx0, ty0, tid);
                __syncthreads();
                tile_level = level;
                wx = inx - tx0; wy = iny - ty0;
                if (STAMPS && P.stamps && stamp_on && tid == 0) atomicAdd(&P.stamps[14], 1ull);   // diagnostic: tiles waited for in place
            }
            LSTAMP(9);
            bilinear_weights(nx - inx, ny - iny, iw00, iw01, iw10, iw11);
            int bpart[2] = {0, 0};
            {
                uint32_t ja[(PP + 1 + 3) / 4], jb[(PP + 1 + 3) / 4];
                lds_row_bytes<PP + 1>(sJ + (wy + r) * SJ_STRIDE, wx + c0, ja);
                lds_row_bytes<PP + 1>(sJ + (wy + r + 1) * SJ_STRIDE, wx + c0, jb);
                const uint32_t W0 = pack_weights(iw00, iw10), W1 = pack_weights(iw01, iw11);
                uint32_t pc = PACKED_COLUMN(ja, jb, 0);
#pragma unroll
                for (int k = 0; k < PP; k += 2) {   // |diff| <= 8160: two of them per register, one dot product per sum and pixel pair
                    const uint32_t p1 = PACKED_COLUMN(ja, jb, k + 1), p2 = PACKED_COLUMN(ja, jb, k + 2);
                    const int diff0 = dot2_acc(p1, W1, dot2_acc(pc, W0, Iv[k])) >> 9;
                    const int diff1 = dot2_acc(p2, W1, dot2_acc(p1, W0, Iv[k + 1])) >> 9;
                    const uint32_t dp = __builtin_amdgcn_perm((uint32_t)diff1, (uint32_t)diff0, 0x05040100u);
                    bpart[0] = dot2_acc(dp, IxP[k >> 1], bpart[0]); bpart[1] = dot2_acc(dp, IyP[k >> 1], bpart[1]);
                    pc = p2;
                }
            }
            LSTAMP(10);
            double sB[2];
            block_sum_exact<2, T>(bpart, sB, sred, slot); slot ^= 1;
            LSTAMP(11);
            const float fb1 = (float)sB[0] * FLT_SCALE, fb2 = (float)sB[1] * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            nx += dx; ny += dy;
            outx = nx + half; outy = ny + half;
            n_iter++;
            if (STAMPS && P.stamps && stamp_on && tid == 0) atomicAdd(&P.stamps[8], 1ull);
            if ((double)dx * dx + (double)dy * dy <= P.eps2d) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                outx -= dx * 0.5f; outy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
            LSTAMP(12);
        }
        if (pre) {   // block-uniform
            __syncthreads();
            tile_commit<T>(sJ, jreg, tid);
            tx0 = ptx0; ty0 = pty0; tile_level = level - 1;
        }
        LSTAMP(4);

        if (status && level == 0) {
            const float fx = outx - half, fy = outy - half;
            const int inx = (int)floorf(fx), iny = (int)floorf(fy);
            if (inx < -W || inx >= lw || iny < -W || iny >= lh) {
                status = 0;
            } else {
                int wx = inx - tx0, wy = iny - ty0;
                if (tile_level != 0 || wx < 0 || wx > 31 || wy < 0 || wy > 31) {
                    tx0 = (inx - 16) & ~3; ty0 = iny - 16;
                    __syncthreads();
                    stage_J<T>(sJ, Jorg, ls, tx0, ty0, tid);
                    __syncthreads();
                    tile_level = 0;
                    wx = inx - tx0; wy = iny - ty0;
                }
                bilinear_weights(fx - inx, fy - iny, iw00, iw01, iw10, iw11);
                int epart[1] = {0};
                uint32_t ja[(PP + 1 + 3) / 4], jb[(PP + 1 + 3) / 4];
                lds_row_bytes<PP + 1>(sJ + (wy + r) * SJ_STRIDE, wx + c0, ja);
                lds_row_bytes<PP + 1>(sJ + (wy + r + 1) * SJ_STRIDE, wx + c0, jb);
                const uint32_t W0 = pack_weights(iw00, iw10), W1 = pack_weights(iw01, iw11);
                uint32_t pc = PACKED_COLUMN(ja, jb, 0);
#pragma unroll
                for (int k = 0; k < PP; k++) {
                    const uint32_t pn = PACKED_COLUMN(ja, jb, k + 1);
                    const int diff = dot2_acc(pn, W1, dot2_acc(pc, W0, Iv[k])) >> 9;
                    epart[0] += diff < 0 ? -diff : diff;
                    pc = pn;
                }
                double sE[1];
                block_sum_exact<1, T>(epart, sE, sred, slot); slot ^= 1;
                err = (float)sE[0] * (1.f / (32 * W * W));
            }
        }
    }
    LSTAMP(5);
    if (STAMPS && P.stamps && stamp_on && tid == 0) atomicAdd(&P.stamps[15], wall_clock64() - wall0);
#undef LSTAMP
    LKResult res;
    res.x = outx; res.y = outy; res.err = err; res.status = status; res.n_iter = n_iter; res.n_lev = n_lev;
    return res;
}

__device__ __forceinline__ void lk_store(const LKResult& r, int t, float* __restrict__ out_xy, uint8_t* __restrict__ out_status,
                                         float* __restrict__ out_err, uint16_t* __restrict__ out_work) {
    if (threadIdx.x == 0) {
        out_xy[2 * t] = r.x; out_xy[2 * t + 1] = r.y;
        out_status[t] = (uint8_t)r.status;
        out_err[t] = r.err;
        if (out_work) out_work[t] = (uint16_t)(r.n_iter | (r.n_lev << 8));   // <= 5 x 30 iterations, <= 5 levels
    }
}

template <bool STAMPS>   // STAMPS: the diagnostic build with phase timers (PMV_LK_STAMPS=1); the product launch carries none of that code
__global__ __launch_bounds__(LK_T) void k_lk(const uint8_t* __restrict__ prevS, const uint8_t* __restrict__ nextS,
                                             PyrLayout L, const float* __restrict__ prev_xy, const int* __restrict__ order, int n, LKParams P,
                                             float* __restrict__ out_xy, uint8_t* __restrict__ out_status,
                                             float* __restrict__ out_err, uint16_t* __restrict__ out_work) {
    const int t = order[blockIdx.x];   // block -> track (XCD-aware order built by the host; -1 = no track)
    if (t < 0 || t >= n) return;
    const LKResult r = lk_track_block<LK_T, STAMPS>(prevS, nextS, L, prev_xy[2 * t], prev_xy[2 * t + 1], P, t == 0);
    lk_store(r, t, out_xy, out_status, out_err, out_work);
}

// Batched form ("same kernels with a leading batch dimension"): the blocks of several independent sequences in one
// launch. seqs[q] = byte offsets of the prev / next frame slots of sequence q inside `slots`; blocks[b] = (q, track) with track
// indexing the concatenated coordinate / result arrays (-1 = padding block). All sequences share the frame geometry L.
constexpr int LKB_T = 64;   // one wavefront per track: throughput form (see block_sum_exact)
template <bool STAMPS>
__global__ __launch_bounds__(LKB_T) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_lk_batch(const uint8_t* __restrict__ slots, const LKBlock* __restrict__ blocks,
                                                   int n_blocks, PyrLayout L, LKParams P, float* __restrict__ out_xy,
                                                   uint8_t* __restrict__ out_status, float* __restrict__ out_err, uint16_t* __restrict__ out_work) {
    // grid-stride over the track list: the launcher may cap the grid (PMV_LK_BATCH_BLOCKS) so that the tracks of a round do not occupy
    // every register-file slot of the chip while the short launches of the back-end chains wait for one
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const LKBlock bk = blocks[b];   // (every record is a real track: the host drops padding entries)
        const int2 bt = make_int2(0, bk.track);
        const LKResult r = lk_track_block<LKB_T, STAMPS>(slots + bk.prev_off, slots + bk.next_off, L, bk.x, bk.y, P, (b & 63) == 17);
        if (STAMPS && P.stamps && (b & 63) == 17 && threadIdx.x == 0) atomicAdd(&P.stamps[13], 1ull);   // diagnostic: sampled tracks
        lk_store(r, bt.y, out_xy, out_status, out_err, out_work);   // (out_work: what this track cost - the next launch's ordering hint, and OPS_lk)
        __syncthreads();
    }
}

hipError_t launch_lk(hipStream_t s, const uint8_t* prev_slot, const uint8_t* next_slot, const PyrLayout& L,
                     const float* d_prev_xy, const int* d_order, int n_blocks, int n, const LKParams& P, float* d_out_xy,
                     uint8_t* d_status, float* d_err, uint16_t* d_work) {
    if (n <= 0) return hipSuccess;
    // every pointer the kernel dereferences: a null base must come back as an error code, never reach a launch
    if (!prev_slot || !next_slot || !d_prev_xy || !d_order || !d_out_xy || !d_status || !d_err || n_blocks < n || L.n_levels < 1 || L.n_levels > MAX_LEVELS) return hipErrorInvalidValue;
    ProfScope ps(K_LK, s);
    if (P.stamps) hipLaunchKernelGGL(k_lk<true>, dim3(n_blocks), dim3(LK_T), 0, s, prev_slot, next_slot, L, d_prev_xy, d_order, n, P, d_out_xy, d_status, d_err, d_work);
    else hipLaunchKernelGGL(k_lk<false>, dim3(n_blocks), dim3(LK_T), 0, s, prev_slot, next_slot, L, d_prev_xy, d_order, n, P, d_out_xy, d_status, d_err, d_work);
    return hipGetLastError();
}

hipError_t launch_lk_batch(hipStream_t s, const uint8_t* slots, const LKBlock* d_blocks, int n_blocks, const PyrLayout& L,
                           const LKParams& P, float* d_out_xy, uint8_t* d_status, float* d_err, uint16_t* d_work) {
    if (n_blocks <= 0) return hipSuccess;
    if (!slots || !d_blocks || !d_out_xy || !d_status || !d_err || L.n_levels < 1 || L.n_levels > MAX_LEVELS) return hipErrorInvalidValue;
    static const int cap = getenv("PMV_LK_BATCH_BLOCKS") ? atoi(getenv("PMV_LK_BATCH_BLOCKS")) : 0;
    // Occupancy cap of the bulk kernel: 13.9 KB of LDS per one-wavefront workgroup lets 11 of them share a CU (160 KB; the prefetch
    // registers of a track, ~150 per lane, allow 12), and then every short kernel of the other classes - the 23 launches of an LM solve,
    // the PnP stages, the detector - waits for LK wavefronts to drain before its workgroups fit anywhere. Unused dynamic LDS on top caps it:
    // 5 000 B = 8 per CU. B = 192, same box, profiles/r03_batch_exp_ae/af.log (after the per-track atomics were gone, which had made
    // the cap look far more valuable than it is: with them every extra resident wavefront was one more contender for three L2 lines):
    // no cap 64.8 k, 2 000 B 63.4 k, 3 000 B 65.6 k, 4 000 B 62.7-68.7 k, 5 000 B 69.0 k, 6 500 B 68.3 k frames/s (run-to-run spread of one
    // setting: +-3 k). PMV_LK_LDS_PAD overrides (0 = no cap).
    static const int lds_pad = getenv("PMV_LK_LDS_PAD") ? atoi(getenv("PMV_LK_LDS_PAD")) : 5000;
    ProfScope ps(K_LK, s);
    const dim3 grid(cap > 0 && cap < n_blocks ? cap : n_blocks);
    const size_t pad = (size_t)(lds_pad > 0 ? lds_pad : 0);
    if (P.stamps) hipLaunchKernelGGL(k_lk_batch<true>, grid, dim3(LKB_T), pad, s, slots, d_blocks, n_blocks, L, P, d_out_xy, d_status, d_err, d_work);
    else hipLaunchKernelGGL(k_lk_batch<false>, grid, dim3(LKB_T), pad, s, slots, d_blocks, n_blocks, L, P, d_out_xy, d_status, d_err, d_work);
    return hipGetLastError();
}

// =========================================================================================================
// goodFeaturesToTrack per grid cell
// =========================================================================================================
__device__ inline unsigned f32_key(float v) {   // order-preserving map float -> uint32
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float f32_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ inline unsigned long long f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double f64_unkey(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

// Phase 1: min-eigenvalue response, 32x32 outputs per 256-thread workgroup.
// grid = (tiles_x, tiles_y, n_cells). cov at the 34x34 halo positions is evaluated at the REFLECT_101-mapped CELL
// coordinate (box filter border = cell), from pixels of the PARENT image with REFLECT_101 on the parent (Sobel border).
__global__ __launch_bounds__(256) void k_gftt_eig(const uint8_t* __restrict__ slots, PyrLayout L,
                                                  const int* __restrict__ cells, float* __restrict__ eig,
                                                  unsigned* __restrict__ cellmax) {
    __shared__ float sC[34 * 34 * 3];
    __shared__ unsigned smax[4];
    const int cell = blockIdx.z;
    const int cx0 = cells[CELL_STRIDE * cell], cy0 = cells[CELL_STRIDE * cell + 1], cw = cells[CELL_STRIDE * cell + 2], ch = cells[CELL_STRIDE * cell + 3];
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
    if (tx0 >= cw || ty0 >= ch) return;
    const uint8_t* img = level_origin(slots + (size_t)cells[CELL_STRIDE * cell + 4] * L.slot_bytes, L, 0);   // the cell's frame slot
    const int st = L.stride[0];
    const float k1 = (float)(1.0 / 3060.0), k2 = (float)(2.0 / 3060.0);
    for (int idx = threadIdx.x; idx < 34 * 34; idx += 256) {
        const int hy = idx / 34, hx = idx - hy * 34;
        const int lx = reflect101(tx0 - 1 + hx, cw), ly = reflect101(ty0 - 1 + hy, ch);
        // padded level 0 == REFLECT_101 of the parent image, so +-1 neighbours are plain loads
        const uint8_t* p = img + (ptrdiff_t)(cy0 + ly) * st + (cx0 + lx);
        const float p00 = p[-st - 1], p01 = p[-st], p02 = p[-st + 1];
        const float p10 = p[-1], p12 = p[1];
        const float p20 = p[st - 1], p21 = p[st], p22 = p[st + 1];
        const float rt = p02 - p00, rm = p12 - p10, rb = p22 - p20;
        const float dx = (rt + rb) * k1 + rm * k2;
        float s_t = k1 * p00; s_t += k2 * p01; s_t += k1 * p02;
        float s_b = k1 * p20; s_b += k2 * p21; s_b += k1 * p22;
        const float dy = s_b - s_t;
        sC[idx * 3 + 0] = dx * dx; sC[idx * 3 + 1] = dx * dy; sC[idx * 3 + 2] = dy * dy;
    }
    __syncthreads();
    unsigned mk = 0;   // key 0 is below every real float key
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
        const int oy = idx >> 5, ox = idx & 31;
        const int x = tx0 + ox, y = ty0 + oy;
        if (x >= cw || y >= ch) continue;
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float* c = &sC[((oy + j) * 34 + (ox + i)) * 3];
                s0 += c[0]; s1 += c[1]; s2 += c[2];
            }
        const float a = (float)s0 * 0.5f, b = (float)s1, c2 = (float)s2 * 0.5f;
        const float e = (a + c2) - sqrtf((a - c2) * (a - c2) + b * b);
        eig[(size_t)cell * CELL_PIX + y * cw + x] = e;
        if (e == e) { const unsigned k = f32_key(e); mk = k > mk ? k : mk; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(mk, o, 64); mk = t > mk ? t : mk; }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mk;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = smax[0];
        for (int i = 1; i < 4; i++) m = smax[i] > m ? smax[i] : m;
        atomicMax(&cellmax[cell], m);
    }
}

// ---- the detector pass proper: no response map in HBM -------------------------------------------------------------------------------
// cov = (dx^2, dx dy, dy^2) at one cell position (the Sobel arithmetic of k_gftt_eig, shared)
__device__ inline void gftt_cov(const uint8_t* __restrict__ p, int st, float& c0, float& c1, float& c2) {
    const float k1 = (float)(1.0 / 3060.0), k2 = (float)(2.0 / 3060.0);
    const float p00 = p[-st - 1], p01 = p[-st], p02 = p[-st + 1];
    const float p10 = p[-1], p12 = p[1];
    const float p20 = p[st - 1], p21 = p[st], p22 = p[st + 1];
    const float rt = p02 - p00, rm = p12 - p10, rb = p22 - p20;
    const float dx = (rt + rb) * k1 + rm * k2;
    float s_t = k1 * p00; s_t += k2 * p01; s_t += k1 * p02;
    float s_b = k1 * p20; s_b += k2 * p21; s_b += k1 * p22;
    const float dy = s_b - s_t;
    c0 = dx * dx; c1 = dx * dy; c2 = dy * dy;
}
// Pass 1, grid = (tiles_x, tiles_y, n_cells), 256 threads, 32x32 cell pixels per workgroup: covariances on the 36x36 halo, min-eigenvalue
// on 34x34, then the 3x3 non-maximum test on the RAW response and one (value, pixel) record per surviving pixel. The reference's order is
// threshold -> dilate -> compare; for a pixel above the threshold the thresholded neighbours exceed it exactly when the raw ones do
// (a neighbour at or below the threshold became 0), so the test needs no threshold - which is only known once the whole cell has been
// seen (0.01 x the cell's maximum, cellinfo[2 cell] by atomicMax). Pass 2 drops the records at or below it. Only values > 0 can ever be
// selected. HBM traffic: the cell's pixels once (+ halo), 8 B per local maximum - no 4 W H map.
__global__ __launch_bounds__(256) void k_gftt_cand(const uint8_t* __restrict__ slots, PyrLayout L, const int* __restrict__ cells,
                                                   float* __restrict__ cand_val, unsigned* __restrict__ cand_idx, unsigned* __restrict__ cellinfo, double quality) {
    __shared__ float sC[36 * 36 * 3];
    __shared__ float sE[34 * 34];
    __shared__ unsigned smax[4];
    BACKEND_PRIO();   // a short pass between two LK launches of its sequence: ahead of the bulk LK waves it shares SIMDs with
    const int cell = blockIdx.z;
    const int cx0 = cells[CELL_STRIDE * cell], cy0 = cells[CELL_STRIDE * cell + 1], cw = cells[CELL_STRIDE * cell + 2], ch = cells[CELL_STRIDE * cell + 3];
    // Consecutive workgroup ids go round the 8 XCDs, each with its own L2: blockIdx.x (8 values = the XCD) is the tile ROW, so the eight
    // tiles that share the 128-byte lines of a 32-row band of the image run on ONE XCD and the band is fetched from HBM once (with
    // blockIdx.x as the tile column every line was fetched by four XCDs: 2.75 MB of HBM reads per 0.47 MB frame by the PMC counters)
    const int tx0 = blockIdx.y * 32, ty0 = blockIdx.x * 32;
    if (tx0 >= cw || ty0 >= ch) return;
    const uint8_t* img = level_origin(slots + (size_t)cells[CELL_STRIDE * cell + 4] * L.slot_bytes, L, 0);
    const int st = L.stride[0];
    for (int idx = threadIdx.x; idx < 36 * 36; idx += 256) {
        const int hy = idx / 36, hx = idx - hy * 36;
        // the box filter's border is the CELL's (REFLECT_101 of the cell coordinate); the Sobel's is the parent image's (= the padded level)
        const int lx = reflect101(tx0 - 2 + hx, cw), ly = reflect101(ty0 - 2 + hy, ch);
        gftt_cov(img + (ptrdiff_t)(cy0 + ly) * st + (cx0 + lx), st, sC[idx * 3], sC[idx * 3 + 1], sC[idx * 3 + 2]);
    }
    __syncthreads();
    unsigned mk = 0;   // key 0 is below every real float key
    for (int idx = threadIdx.x; idx < 34 * 34; idx += 256) {
        const int ey = idx / 34, ex = idx - ey * 34;
        const int x = tx0 - 1 + ex, y = ty0 - 1 + ey;
        float e = -1.f;   // outside the cell: never consulted (only interior pixels are tested), never counted in the maximum
        if (x >= 0 && y >= 0 && x < cw && y < ch) {
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const float* c = &sC[((ey + j) * 36 + (ex + i)) * 3];
                    s0 += c[0]; s1 += c[1]; s2 += c[2];
                }
            const float a = (float)s0 * 0.5f, b = (float)s1, c2 = (float)s2 * 0.5f;
            e = (a + c2) - sqrtf((a - c2) * (a - c2) + b * b);
            if (ex >= 1 && ex <= 32 && ey >= 1 && ey <= 32 && e == e) { const unsigned k = f32_key(e); mk = k > mk ? k : mk; }   // every cell pixel once
        }
        sE[idx] = e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(mk, o, 64); mk = t > mk ? t : mk; }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mk;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = smax[0];
        for (int i = 1; i < 4; i++) m = smax[i] > m ? smax[i] : m;
        atomicMax(&cellinfo[2 * cell], m);
    }
    // the tile's records are collected in LDS first (sC is free now), then ONE global atomic per workgroup reserves their place in the
    // cell's list (640 workgroups bumping 10 counters per wavefront and turn serialised in L2: 82 us per launch instead of 15)
    // A record at or below quality x (the largest response seen so far) can be dropped here already: the cell's maximum only grows, the
    // threshold with it (the same (float)((double)max * quality) as pass 2 uses, monotone in max), so pass 2 would drop it anyway. "So far" =
    // this tile's own maximum (reading what the other tiles have published - one more global load per thread behind the atomicMax - cost
    // 15 us per launch, measured): fewer records written, read back and compacted, at no cost.
    unsigned seen = smax[0];
    for (int i = 1; i < 4; i++) seen = smax[i] > seen ? smax[i] : seen;
    const float thr_now = seen ? (float)((double)f32_unkey(seen) * quality) : 0.f;
    unsigned* s_cnt = smax;                          // [0] tile count, [1] base in the cell's list
    float* lv = sC;                                  // up to 1024 values ...
    unsigned* li = (unsigned*)(sC + 1024);           // ... and pixel indices
    __syncthreads();
    if (threadIdx.x == 0) s_cnt[0] = 0;
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {   // (every lane takes all four turns: the ballot below is wave-wide)
        const int oy = idx >> 5, ox = idx & 31;
        const int x = tx0 + ox, y = ty0 + oy;
        bool cand = false;
        float v = 0.f;
        if (x >= 1 && y >= 1 && x < cw - 1 && y < ch - 1) {
            v = sE[(oy + 1) * 34 + ox + 1];
            cand = v > 0.f && v > thr_now;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) cand = cand && !(sE[(oy + j) * 34 + ox + i] > v);
        }
        const unsigned long long bal = __ballot(cand);
        if (bal) {   // one LDS atomic per wavefront
            const int lane = threadIdx.x & 63;
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[0], (unsigned)__popcll(bal));
            base = __shfl(base, 0, 64);
            if (cand) {
                const unsigned slot = base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                lv[slot] = v; li[slot] = (unsigned)(y * cw + x);
            }
        }
    }
    __syncthreads();
    const unsigned nt = s_cnt[0];
    if (nt == 0) return;
    if (threadIdx.x == 0) s_cnt[1] = atomicAdd(&cellinfo[2 * cell + 1], nt);
    __syncthreads();
    const unsigned gbase = s_cnt[1];
    float* cv = cand_val + (size_t)cell * CELL_PIX + gbase;
    unsigned* ci = cand_idx + (size_t)cell * CELL_PIX + gbase;
    for (unsigned i = threadIdx.x; i < nt; i += 256) { cv[i] = lv[i]; ci[i] = li[i]; }
}

// Pass 2, one 256-thread workgroup per cell: records above the threshold are compacted into LDS and dealt to the lanes' REGISTERS; the
// greedy selection of cv::goodFeaturesToTrack ("walk the list sorted by (value, address) descending, accept a corner unless an accepted
// one is closer than minDistance") runs as: repeat {arg-max over the live records; accept; kill every record closer than minDistance} -
// identical result, max_corners rounds of register work with one 4-wavefront exchange each, no sort. A cell with more than GP_REG records
// above the threshold (a periodic texture can make every pixel a local maximum) takes the same rounds over its list in HBM.
// Four wavefronts per cell. Measured alternatives (shader-clock stamps, PMV_GFTT_DBG=1; one cell = 2740 raw records, 742 above the threshold):
// a round is a serial chain (arg-max -> accept -> kill) and costs ~2.9 k cycles whatever the operand width (64-bit keys: 3.4 k) - a lone or
// nearly lone wavefront issues one DEPENDENT instruction every 10-30 cycles, so the chain's instruction count is what matters; ONE wavefront
// per cell (no exchange, no barrier, 12 slots per lane) takes 9.6 k cycles per round: the slots' instructions do not overlap, they queue.
// SORTING the records once (bitonic network over the LDS list, 55 stages for 1024 records) and letting one wavefront walk the sorted list 64
// records at a time (accept in list order unless an accepted corner is closer than minDistance: OpenCV's loop verbatim, bit-exact in every
// test) took 84 us per launch instead of 60: a stage with its barrier costs ~1 us, not the 200 cycles of its instruction count.
constexpr int GP_T = 256, GP_SLOTS = 16, GP_REG = GP_T * GP_SLOTS, GP_MAXOUT = 4096;   // GP_MAXOUT = MAX_PER_CELL of pmv_ctx.h
__global__ __launch_bounds__(GP_T) void k_gftt_pick(const int* __restrict__ cells, float* __restrict__ cand_val, unsigned* __restrict__ cand_idx,
                                                    const unsigned* __restrict__ cellinfo, int max_corners, double quality, double min_dist,
                                                    int unlimited, int* __restrict__ out_xy, int* __restrict__ out_count, int* __restrict__ flags) {
    __shared__ unsigned long long lkey[GP_REG];
    __shared__ __attribute__((aligned(16))) unsigned long long wbest[2][4];
    // accepted corners, (y << 16) | x, written to HBM once at the end: a global store per round sits in front of the next round's barrier
    // (__syncthreads waits for the wavefront's outstanding stores: 1-2 us each - 80 of the 93 us this kernel took with the store inside the loop)
    __shared__ unsigned s_acc[GP_MAXOUT];
    BACKEND_PRIO();
    const unsigned long long dbg_t0 = __builtin_readcyclecounter();
    const int cell = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = cells[CELL_STRIDE * cell + 2];
    float* cv = cand_val + (size_t)cell * CELL_PIX;
    unsigned* ci = cand_idx + (size_t)cell * CELL_PIX;
    const int nraw = (int)cellinfo[2 * cell + 1];
    const double maxVal = (double)f32_unkey(cellinfo[2 * cell]);
    const float thr = (float)(maxVal * quality);
    // compaction of the records above the threshold: key = (value bits, pixel index), 0 = dead. The first GP_REG go to LDS, the rest
    // stay in the HBM list, compacted in place (a record only moves to a lower position, behind every record already read)
    int n = 0;
    constexpr int CPT = 4;   // records per thread and chunk: a quarter of the barriers
    for (int base = 0; base < nraw; base += GP_T * CPT) {
        float v[CPT]; unsigned px[CPT]; bool keep[CPT];
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            const int i = base + q * GP_T + tid;
            v[q] = 0.f; px[q] = 0;
            if (i < nraw) { v[q] = cv[i]; px[q] = ci[i]; }
            keep[q] = i < nraw && v[q] > thr;
        }
        __syncthreads();   // (all reads of this chunk before any in-place write into it)
        // place in the compacted list = records kept so far + those of the wavefronts before + those of this wavefront's earlier turns + rank
        // in the ballot (any order will do, the selection is order-free; every target position lies inside the part already read)
        int wave_tot = 0;
#pragma unroll
        for (int q = 0; q < CPT; q++) wave_tot += __popcll(__ballot(keep[q]));
        int off = n, tot = wave_tot;
        if (GP_T > 64) {
            if (lane == 0) wbest[0][wave] = (unsigned long long)wave_tot;
            __syncthreads();
            for (int w2 = 0; w2 < wave; w2++) off += (int)wbest[0][w2];
            tot = (int)(wbest[0][0] + wbest[0][1] + wbest[0][2] + wbest[0][3]);
        }
        int run = 0;
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            const unsigned long long bal = __ballot(keep[q]);
            if (keep[q]) {
                const int slot = off + run + __popcll(bal & ((1ull << lane) - 1ull));
                if (slot < GP_REG) lkey[slot] = ((unsigned long long)__float_as_uint(v[q]) << 32) | px[q];
                else { cv[slot] = v[q]; ci[slot] = px[q]; }
            }
            run += __popcll(bal);
        }
        n += tot;
    }
    __syncthreads();
    const unsigned long long dbg_t1 = __builtin_readcyclecounter();
    const bool use_dist = min_dist >= 1.0;
    const double md2 = min_dist * min_dist;
    // dx*dx + dy*dy is an integer: (double)d2 < md2  <=>  d2 < ceil(md2)
    const int md2i = (int)(md2 < 2.0e9 ? ceil(md2) : 2.0e9);
    const int n_reg = n < GP_REG ? n : GP_REG;
    // A record in registers = (value bits, (y << 16) | x): the pair orders like (value, pixel index = y * cw + x), the coordinates need no
    // division by the cell width in the rounds, and every comparison is a full-rate 32-bit one (v_cmp / v_cndmask on 64-bit keys are not)
    unsigned kv[GP_SLOTS], kp[GP_SLOTS];   // kv == 0: dead / empty (live values are > 0)
#pragma unroll
    for (int s2 = 0; s2 < GP_SLOTS; s2++) {
        const int i = s2 * GP_T + tid;
        const unsigned long long k = i < n_reg ? lkey[i] : 0ull;
        kv[s2] = (unsigned)(k >> 32);
        const int px = (int)(k & 0x7fffffffu);
        const int y = px / cw;
        kp[s2] = ((unsigned)y << 16) | (unsigned)(px - y * cw);
    }
    unsigned* wv = (unsigned*)&wbest[0][0];   // [2][4] values, then [2][4] positions
    unsigned* wp = wv + 8;
    int naccepted = 0;
    const unsigned long long dbg_t2 = __builtin_readcyclecounter();
    const int nslots = (n_reg + GP_T - 1) / GP_T;   // slots in use (uniform): a typical cell keeps ~750 records = 3 of the 16
    unsigned bv = 0, bp = 0;                        // this lane's best live record; recomputed only after one of its records died
#pragma unroll
    for (int s2 = 0; s2 < GP_SLOTS; s2++) { const bool g = kv[s2] > bv || (kv[s2] == bv && kp[s2] > bp); bv = g ? kv[s2] : bv; bp = g ? kp[s2] : bp; }
    for (int it = 0; it < max_corners; it++) {
        unsigned tv = bv, tp = bp;
        for (int i = GP_REG + tid; i < n; i += GP_T) {   // (only for cells beyond the register capacity: records in HBM, bit 31 of the index = dead)
            const unsigned px = ci[i];
            if (px >> 31) continue;
            const unsigned v = __float_as_uint(cv[i]);
            const int y = (int)px / cw;
            const unsigned pp = ((unsigned)y << 16) | (unsigned)((int)px - y * cw);
            const bool g = v > tv || (v == tv && pp > tp);
            tv = g ? v : tv; tp = g ? pp : tp;
        }
        // arg-max over the workgroup: first the value, then the position among the lanes that hold that value
        const unsigned mv = wave_max_u32(tv);
        // the position: almost always ONE lane holds the maximum value - read its position directly; a second reduction only on a tie
        const unsigned long long holders = __ballot(tv == mv);
        unsigned mp;
        if (__popcll(holders) == 1) mp = (unsigned)__builtin_amdgcn_readlane((int)tp, __builtin_ctzll(holders));
        else mp = wave_max_u32(tv == mv ? tp : 0u);
        unsigned V = mv, P = mp;
        if (GP_T > 64) {   // several wavefronts per cell: one exchange through LDS per round (the buffer alternates)
            if (lane == 0) { wv[(it & 1) * 4 + wave] = mv; wp[(it & 1) * 4 + wave] = mp; }
            __syncthreads();
            const uint4 v4 = *(const uint4*)&wv[(it & 1) * 4], p4 = *(const uint4*)&wp[(it & 1) * 4];
            V = max(max(v4.x, v4.y), max(v4.z, v4.w));
            P = max(max(v4.x == V ? p4.x : 0u, v4.y == V ? p4.y : 0u), max(v4.z == V ? p4.z : 0u, v4.w == V ? p4.w : 0u));
        }
        if (V == 0) break;
        const int bx = (int)(P & 0xffffu), by = (int)(P >> 16);
        if (tid == 0) s_acc[naccepted] = P;
        naccepted++;
        bool died = false;
#pragma unroll
        for (int s2 = 0; s2 < GP_SLOTS; s2++) {
            if (s2 < nslots) {   // uniform (no break: the loop has to unroll, the slots are registers)
                bool kill;
                if (use_dist) {
                    const int dx = (int)(kp[s2] & 0xffffu) - bx, dy = (int)(kp[s2] >> 16) - by;
                    kill = dx * dx + dy * dy < md2i;
                } else kill = kp[s2] == P;
                kill = kill && kv[s2] != 0;
                if (kill) kv[s2] = 0;
                died = died || kill;
            }
        }
        if (died) {
            bv = 0; bp = 0;
#pragma unroll
            for (int s2 = 0; s2 < GP_SLOTS; s2++) {
                if (s2 < nslots) {
                    const bool g = kv[s2] > bv || (kv[s2] == bv && kp[s2] > bp);
                    bv = g ? kv[s2] : bv; bp = g ? kp[s2] : bp;
                }
            }
        }
        for (int i = GP_REG + tid; i < n; i += GP_T) {
            const unsigned u = ci[i];
            if (u >> 31) continue;
            bool kill;
            const int y = (int)u / cw, x = (int)u - y * cw;
            if (use_dist) {
                const int dx = x - bx, dy = y - by;
                kill = dx * dx + dy * dy < md2i;
            } else kill = x == bx && y == by;
            if (kill) ci[i] = u | 0x80000000u;
        }
    }
    // cv::goodFeaturesToTrack(maxCorners <= 0) has no limit; here the caller's buffer holds max_corners: more corners than that
    // is reported (bit 2), not silently truncated
    if (unlimited && naccepted == max_corners) {
        int live = 0;
#pragma unroll
        for (int s2 = 0; s2 < GP_SLOTS; s2++) live |= kv[s2] != 0;
        for (int i = GP_REG + tid; i < n; i += GP_T) live |= !(ci[i] >> 31);
        if (live) atomicOr(flags, 4);
    }
    if (cell == 0 && tid == 0 && (flags[0] & 0x40000000)) {   // diagnostic (PMV_GFTT_DBG=1): cycles of compaction / key set-up / rounds, record counts
        const unsigned long long dbg_t3 = __builtin_readcyclecounter();
        flags[1] = (int)(dbg_t1 - dbg_t0); flags[2] = (int)(dbg_t2 - dbg_t1); flags[3] = (int)(dbg_t3 - dbg_t2);
        flags[0] = 0x40000000 | (nraw & 0xffff) | ((n & 0x3fff) << 16);
    }
    __syncthreads();
    for (int i = tid; i < naccepted; i += GP_T) {
        const unsigned a = s_acc[i];   // (y << 16) | x
        *(int2*)&out_xy[((size_t)cell * max_corners + i) * 2] = make_int2((int)(a & 0xffffu), (int)(a >> 16));
    }
    if (tid == 0) out_count[cell] = naccepted;
}

hipError_t launch_gftt(hipStream_t s, const uint8_t* slots, const PyrLayout& L, const int* d_cells, int n_cells,
                       int max_per_cell, double quality, double min_dist, int unlimited, float* d_eig, unsigned* d_cellmax,
                       int* d_out_xy, int* d_out_count, int* d_flags, unsigned* d_spill) {
    if (!slots || !d_cells || !d_eig || !d_cellmax || !d_out_xy || !d_out_count || !d_flags || !d_spill || n_cells < 1 || max_per_cell < 1 || max_per_cell > GP_MAXOUT) return hipErrorInvalidValue;
    // d_eig / d_spill: the cells' candidate records (value / pixel index, CELL_PIX each); d_cellmax: (maximum key, record count) per cell
    hipError_t e = hipMemsetAsync(d_cellmax, 0, 2 * sizeof(unsigned) * n_cells, s);
    if (e != hipSuccess) return e;
    { ProfScope ps(K_GFTT_CAND, s);
    hipLaunchKernelGGL(k_gftt_cand, dim3(8, 8, n_cells), dim3(256), 0, s, slots, L, d_cells, d_eig, d_spill, d_cellmax, quality); }
    ProfScope ps2(K_GFTT_PICK, s);
    hipLaunchKernelGGL(k_gftt_pick, dim3(n_cells), dim3(GP_T), 0, s, d_cells, d_eig, d_spill, d_cellmax, max_per_cell,
                       quality, min_dist, unlimited, d_out_xy, d_out_count, d_flags);
    return hipGetLastError();
}
// diagnostic / parity: the response map itself (pmv_debug_gftt_response), which the detector pass no longer writes
hipError_t launch_gftt_response(hipStream_t s, const uint8_t* slots, const PyrLayout& L, const int* d_cells, int n_cells, float* d_eig, unsigned* d_cellmax) {
    if (!slots || !d_cells || !d_eig || !d_cellmax || n_cells < 1) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_cellmax, 0, 2 * sizeof(unsigned) * n_cells, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gftt_eig, dim3(8, 8, n_cells), dim3(256), 0, s, slots, L, d_cells, d_eig, d_cellmax);
    return hipGetLastError();
}

// =========================================================================================================
// ShiTomasiFeatureExtractor (in-repo arithmetic, float64)
// =========================================================================================================
__global__ __launch_bounds__(256) void k_st_resp(const uint8_t* __restrict__ slots, PyrLayout L,
                                                 const int* __restrict__ cells, double* __restrict__ resp,
                                                 unsigned long long* __restrict__ cellmax) {
    __shared__ double sH[34 * 34 * 3];
    __shared__ unsigned long long smax[4];
    const int cell = blockIdx.z;
    const int cx0 = cells[CELL_STRIDE * cell], cy0 = cells[CELL_STRIDE * cell + 1], cw = cells[CELL_STRIDE * cell + 2], ch = cells[CELL_STRIDE * cell + 3];
    const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
    if (tx0 >= cw || ty0 >= ch) return;
    const uint8_t* img = level_origin(slots + (size_t)cells[CELL_STRIDE * cell + 4] * L.slot_bytes, L, 0);
    const int st = L.stride[0];
    for (int idx = threadIdx.x; idx < 34 * 34; idx += 256) {
        const int hy = idx / 34, hx = idx - hy * 34;
        const int lx = reflect101(tx0 - 1 + hx, cw), ly = reflect101(ty0 - 1 + hy, ch);
        double gx = 0.0, gy = 0.0;   // Frame.cpp:58-86: zero on the cell border
        if (lx >= 1 && lx < cw - 1 && ly >= 1 && ly < ch - 1) {
            const uint8_t* p = img + (ptrdiff_t)(cy0 + ly) * st + (cx0 + lx);
            gx = 1. / 2. * (double)(int8_t)p[1] - 1. / 2. * (double)(int8_t)p[-1];     // quirk Q1: signed char view
            gy = 1. / 2. * (double)(int8_t)p[st] - 1. / 2. * (double)(int8_t)p[-st];
        }
        sH[idx * 3 + 0] = gx * gx; sH[idx * 3 + 1] = gy * gy; sH[idx * 3 + 2] = gx * gy;
    }
    __syncthreads();
    unsigned long long mk = 0;
    const double inv9 = 1.0 / 9.0;
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
        const int oy = idx >> 5, ox = idx & 31;
        const int x = tx0 + ox, y = ty0 + oy;
        if (x >= cw || y >= ch) continue;
        double R = 0.0;
        if (x < cw - 1) {   // ShiTomasiFeatureExtractor.cpp:58 skips the last column
            double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double* h = &sH[((oy + j) * 34 + (ox + i)) * 3];
                    s0 += h[0]; s1 += h[1]; s2 += h[2];
                }
            const double Ixx = s0 * inv9, Iyy = s1 * inv9, Ixy = s2 * inv9;
            const double B = -Ixx - Iyy;
            const double C = Ixx * Iyy - Ixy * Ixy;
            const double disc = sqrt(B * B - 4 * C);
            const double l1 = (-B + disc) / 2, l2 = (-B - disc) / 2;
            R = (l2 < l1) ? l2 : l1;   // std::min(l1, l2)
        }
        resp[(size_t)cell * CELL_PIX + y * cw + x] = R;
        if (R == R) { const unsigned long long k = f64_key(R); mk = k > mk ? k : mk; }
    }
    mk = wave_max_u64(mk);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mk;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = smax[0];
        for (int i = 1; i < 4; i++) m = smax[i] > m ? smax[i] : m;
        atomicMax(&cellmax[cell], m);
    }
}

constexpr int ST_CAP = 8192;
constexpr size_t ST_SELECT_SHM = ST_CAP * 12 + 16 * 12 + 16;
__global__ __launch_bounds__(1024) void k_st_select(const int* __restrict__ cells, const double* __restrict__ resp,
                                                    const unsigned long long* __restrict__ cellmax, int max_feats,
                                                    double quality, int* __restrict__ out_xy,
                                                    double* __restrict__ out_score, int* __restrict__ out_count,
                                                    int* __restrict__ flags, unsigned* __restrict__ spill_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* spill = spill_all + (size_t)blockIdx.x * CELL_PIX;   // candidates beyond ST_CAP: pixel indices in HBM (bit 31 = taken), see k_gftt_select
    unsigned long long* ckey = (unsigned long long*)smem;              // ST_CAP: ordered score key, 0 = dead
    unsigned* cidx = (unsigned*)(smem + ST_CAP * 8);                   // ST_CAP
    unsigned long long* wk = (unsigned long long*)(smem + ST_CAP * 12);// 16
    unsigned* wi = (unsigned*)(smem + ST_CAP * 12 + 16 * 8);           // 16
    int* scount = (int*)(smem + ST_CAP * 12 + 16 * 12);
    const int cell = blockIdx.x;
    const int cw = cells[CELL_STRIDE * cell + 2], ch = cells[CELL_STRIDE * cell + 3];
    const double* R = resp + (size_t)cell * CELL_PIX;
    const int tid = threadIdx.x;
    if (tid == 0) *scount = 0;
    __syncthreads();
    const double rmax = f64_unkey(cellmax[cell]);
    const double thr = rmax * quality;
    for (int idx = tid; idx < cw * ch; idx += 1024) {
        const double v = R[idx];
        if (v > thr) {
            const int slot = atomicAdd(scount, 1);
            if (slot < ST_CAP) { ckey[slot] = f64_key(v); cidx[slot] = (unsigned)idx; }
            else spill[slot - ST_CAP] = (unsigned)idx;
        }
    }
    __threadfence_block();
    __syncthreads();
    const int ncand = *scount;
    int nacc = 0;
    for (int it = 0; it < max_feats; it++) {
        // best = highest score, ties -> lowest raster index (stable sort of a raster-ordered list)
        unsigned long long bk = 0; unsigned bi = 0xffffffffu;
        for (int i = tid; i < ncand; i += 1024) {
            unsigned long long k; unsigned ci;
            if (i < ST_CAP) { k = ckey[i]; ci = cidx[i]; }
            else { ci = spill[i - ST_CAP]; k = (ci >> 31) ? 0ull : f64_key(R[ci]); }
            if (k > bk || (k == bk && k != 0 && ci < bi)) { bk = k; bi = ci; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long tk = __shfl_xor(bk, o, 64);
            const unsigned ti = __shfl_xor(bi, o, 64);
            if (tk > bk || (tk == bk && ti < bi)) { bk = tk; bi = ti; }
        }
        if ((tid & 63) == 0) { wk[tid >> 6] = bk; wi[tid >> 6] = bi; }
        __syncthreads();
        bk = wk[0]; bi = wi[0];
#pragma unroll
        for (int i = 1; i < 16; i++)
            if (wk[i] > bk || (wk[i] == bk && wi[i] < bi)) { bk = wk[i]; bi = wi[i]; }
        if (bk == 0) break;
        if (tid == 0) {
            out_xy[((size_t)cell * max_feats + nacc) * 2] = (int)(bi % (unsigned)cw);
            out_xy[((size_t)cell * max_feats + nacc) * 2 + 1] = (int)(bi / (unsigned)cw);
            out_score[(size_t)cell * max_feats + nacc] = f64_unkey(bk);
        }
        nacc++;
        for (int i = tid; i < ncand; i += 1024) {
            if (i < ST_CAP) { if (cidx[i] == bi) ckey[i] = 0; }
            else if (spill[i - ST_CAP] == bi) spill[i - ST_CAP] = bi | 0x80000000u;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) out_count[cell] = nacc;
}

hipError_t launch_shitomasi(hipStream_t s, const uint8_t* slots, const PyrLayout& L, const int* d_cells, int n_cells,
                            int max_per_cell, double quality, double* d_resp, unsigned long long* d_cellmax,
                            int* d_out_xy, double* d_out_score, int* d_out_count, int* d_flags, unsigned* d_spill) {
    if (!slots || !d_cells || !d_resp || !d_cellmax || !d_out_xy || !d_out_score || !d_out_count || !d_flags || !d_spill || n_cells < 1 || max_per_cell < 1) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_cellmax, 0, sizeof(unsigned long long) * n_cells, s);
    if (e != hipSuccess) return e;
    { ProfScope ps(K_ST_RESP, s);
    hipLaunchKernelGGL(k_st_resp, dim3(8, 8, n_cells), dim3(256), 0, s, slots, L, d_cells, d_resp, d_cellmax); }
    ProfScope ps2(K_ST_SELECT, s);
    const size_t shm = ST_SELECT_SHM;
    hipLaunchKernelGGL(k_st_select, dim3(n_cells), dim3(1024), shm, s, d_cells, d_resp, d_cellmax, max_per_cell,
                       quality, d_out_xy, d_out_score, d_out_count, d_flags, d_spill);
    return hipGetLastError();
}

// Kernels that need more than the default 64 KB of dynamic LDS opt in per DEVICE (function attributes are per device in HIP):
// called by pmv_ctx_create after hipSetDevice, so a second context on another GPU of the same process is set up as well.
hipError_t frontend_prepare_device() {
    return hipFuncSetAttribute((const void*)k_st_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ST_SELECT_SHM);
}

}  // namespace pmv
