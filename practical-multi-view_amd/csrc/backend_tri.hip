// gfx950 two-view triangulation for the reference's triangulator (OpenCVFivePointTri.cpp:27, cv::recoverPose): for each of
// the four (R, t) candidates of the essential matrix, every correspondence is DLT-triangulated (cvTriangulatePoints: the
// eigenvector of the smallest eigenvalue of the 4x4 A^T A, cyclic Jacobi) and run through the cheirality tests. One thread
// per (candidate, point); same operation order as host/vo_fivepoint.cpp:dlt_candidates_host => bit-identical results.
#include "pmv_ctx.h"
#include "backend.h"
#include "pmv_prof.h"
#include "pmv_dense.h"

namespace pmv {

__device__ __forceinline__ void tri_dlt_body(const double* __restrict__ P1x4, const double* __restrict__ q1,
                                             const double* __restrict__ q2, const uint8_t* __restrict__ mask_in, int n,
                                             double* __restrict__ Q, uint8_t* __restrict__ mask, const int i, const int c) {
    if (i >= n) return;
    double P1[12];
#pragma unroll
    for (int k = 0; k < 12; k++) P1[k] = P1x4[c * 12 + k];
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double x0 = q1[2 * i], y0 = q1[2 * i + 1], x1 = q2[2 * i], y1 = q2[2 * i + 1];
    double A[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        A[0 * 4 + k] = x0 * P0[8 + k] - P0[k];
        A[1 * 4 + k] = y0 * P0[8 + k] - P0[4 + k];
        A[2 * 4 + k] = x1 * P1[8 + k] - P1[k];
        A[3 * 4 + k] = y1 * P1[8 + k] - P1[4 + k];
    }
    double AtA[16], w4[4], V4[16];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) acc += A[k * 4 + a] * A[k * 4 + b];
            AtA[a * 4 + b] = acc;
        }
    d_jacobi_eig<4>(AtA, w4, V4);
    const double Qh[4] = {V4[0], V4[4], V4[8], V4[12]};
#pragma unroll
    for (int k = 0; k < 4; k++) Q[((size_t)c * 4 + k) * n + i] = Qh[k];
    const double huge = __builtin_huge_val();
    bool m = Qh[2] * Qh[3] > 0;
    const double Qn[4] = {Qh[0] / Qh[3], Qh[1] / Qh[3], Qh[2] / Qh[3], Qh[3] / Qh[3]};
    m = m && (Qn[2] < huge);
    const double z2 = P1[8] * Qn[0] + P1[9] * Qn[1] + P1[10] * Qn[2] + P1[11] * Qn[3];
    m = m && (z2 > 0) && (z2 < huge);
    m = m && (mask_in[i] != 0);
    mask[(size_t)c * n + i] = m ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_tri_dlt(const double* __restrict__ P1x4, const double* __restrict__ q1,
                                                const double* __restrict__ q2, const uint8_t* __restrict__ mask_in, int n,
                                                double* __restrict__ Q, uint8_t* __restrict__ mask) { BACKEND_PRIO();
    tri_dlt_body(P1x4, q1, q2, mask_in, n, Q, mask, blockIdx.x * 64 + threadIdx.x, blockIdx.y);
}
// batched: blockIdx.z = problem
__global__ __launch_bounds__(64) void k_tri_dlt_batch(const DltProblem* __restrict__ probs) { BACKEND_PRIO();
    const DltProblem p = probs[blockIdx.z];
    tri_dlt_body(p.P1x4, p.q1, p.q2, p.mask_in, p.n, p.Q, p.mask, blockIdx.x * 64 + threadIdx.x, blockIdx.y);
}
hipError_t launch_tri_dlt_batch(hipStream_t s, const DltProblem* d_probs, int n_probs, int max_n) {
    if (n_probs <= 0 || max_n <= 0) return hipSuccess;
    if (!d_probs) return hipErrorInvalidValue;
    ProfScope ps(K_TRI_DLT, s);
    hipLaunchKernelGGL(k_tri_dlt_batch, dim3((max_n + 63) / 64, 4, n_probs), dim3(64), 0, s, d_probs);
    return hipGetLastError();
}

hipError_t launch_tri_dlt(hipStream_t s, const double* d_P1x4, const double* d_q1, const double* d_q2, const uint8_t* d_mask_in, int n,
                          double* d_Q, uint8_t* d_mask) {
    ProfScope ps(K_TRI_DLT, s);
    hipLaunchKernelGGL(k_tri_dlt, dim3((n + 63) / 64, 4), dim3(64), 0, s, d_P1x4, d_q1, d_q2, d_mask_in, n, d_Q, d_mask);
    return hipGetLastError();
}

}  // namespace pmv
