// gfx950 PnP: batched RANSAC hypotheses (5-point EPnP, one wavefront per hypothesis, which also scores its model on all
// points), the sequential RANSAC bookkeeping (adaptive iteration cut-off) replayed on the device, and the Levenberg–Marquardt
// refit on the inliers — everything cv::solvePnPRansac does behind /root/reference/OpenCVEPnPSolver.cpp:35-36.
// The sample index stream (cv::RNG((uint64)-1), 5 distinct indices per iteration) depends only on the point count, so the
// host adapter generates all `iterations` samples up front; evaluating them in parallel and then replaying the
// "best so far / niters = RANSACUpdateNumIters(...)" scan in iteration order gives exactly the sequential result.
#include "pmv_ctx.h"
#include "backend.h"
#include "pmv_prof.h"
#include "pmv_dense.h"
#include <float.h>

namespace pmv {

#define WAVE_SYNC_PNP() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ---- EPnP on 5 points; per-hypothesis state lives in LDS (one wavefront = one hypothesis) ---------------------------------
struct EpnpShared {
    double A[144], V[144];
    double M[10 * 12];
    double C[6], S[6];
    int P[6], Q[6];
    double pws[15], us[10], alphas[20], cws[12];
    double v4[48];
    double L[60], rho[6];
    double case_err[3], case_R[27], case_t[9];
};

// epnp::qr_solve (6x4, Householder), same arithmetic in the same order as the pointer-walking original — including its
// pivot-scale quirk: eta is the maximum over rows k..4 (the scan starts one element early and never reads the last row).
// Every index is a compile-time constant after unrolling, so A, b, A1, A2 live in registers (the pointer version kept A in
// scratch memory: 70 scratch accesses per call on the critical path).
__device__ inline void epnp_qr_solve(double* A, double* b, double* X) {
    constexpr int nr = 6, nc = 4;
    double A1[4], A2[4];
#pragma unroll
    for (int k = 0; k < nc; k++) {
        double eta = fabs(A[k * nc + k]);
#pragma unroll
        for (int i = k + 1; i < nr; i++) {
            const double elt = fabs(A[(i - 1) * nc + k]);
            if (eta < elt) eta = elt;
        }
        if (eta == 0) { X[0] = X[1] = X[2] = X[3] = 0.0; return; }
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
#pragma unroll
        for (int i = k; i < nr; i++) { A[i * nc + k] *= inv_eta; sum2 += A[i * nc + k] * A[i * nc + k]; }
        double sigma = sqrt(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
#pragma unroll
        for (int j = k + 1; j < nc; j++) {
            double sum = 0;
#pragma unroll
            for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
            const double tau = sum / A1[k];
#pragma unroll
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
#pragma unroll
    for (int j = 0; j < nc; j++) {
        double tau = 0;
#pragma unroll
        for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
#pragma unroll
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
#pragma unroll
    for (int i = nc - 2; i >= 0; i--) {
        double sum = 0;
#pragma unroll
        for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

__device__ inline void epnp_gauss_newton(const double* L, const double* rho, double* betas) {
    for (int k = 0; k < 5; k++) {
        double A[24], b[6], x[4];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double* rowL = L + i * 10;
            double* rowA = A + i * 4;
            rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
            rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
            rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
            rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
            b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                             rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                             rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                             rowL[9] * betas[3] * betas[3]);
        }
        epnp_qr_solve(A, b, x);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
}

// compute_R_and_t for one beta set (n = 5 points); returns the mean reprojection error
__device__ double epnp_R_and_t(const EpnpShared& sh, const double* betas, double fu, double fv, double uc, double vc,
                               double R[9], double t[3]) {
    const int n = 5;
    double ccs[12], pcs[15];
    for (int i = 0; i < 12; i++) ccs[i] = 0.0;
    for (int i = 0; i < 4; i++) {
        const double* v = sh.v4 + 12 * i;
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) ccs[j * 3 + k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
        const double* a = &sh.alphas[4 * i];
        for (int j = 0; j < 3; j++) pcs[3 * i + j] = a[0] * ccs[j] + a[1] * ccs[3 + j] + a[2] * ccs[6 + j] + a[3] * ccs[9 + j];
    }
    if (pcs[2] < 0.0) {
        for (int i = 0; i < 12; i++) ccs[i] = -ccs[i];
        for (int i = 0; i < 15; i++) pcs[i] = -pcs[i];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += sh.pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double* pc = &pcs[3 * i];
        const double* pw = &sh.pws[3 * i];
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    double U[9], s[3], V[9];
    d_svd3(abt, U, s, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    t[0] = pc0[0] - d_dot3(R, pw0);
    t[1] = pc0[1] - d_dot3(R + 3, pw0);
    t[2] = pc0[2] - d_dot3(R + 6, pw0);
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
        const double* pw = &sh.pws[3 * i];
        const double Xc = d_dot3(R, pw) + t[0], Yc = d_dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (d_dot3(R + 6, pw) + t[2]);
        const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
        const double u = sh.us[2 * i], v = sh.us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

// hypothesis h of one RANSAC problem, executed by one wavefront. models: n_hyp x 6 (rvec, tvec)
// LDS_L: the Gauss-Newton passes read L_6x10 / rho from LDS instead of a register copy - 130 fewer registers (two wavefronts per SIMD
// instead of one) for a longer chain; the throughput form used by the batched launch. Same expressions either way.
template <bool LDS_L>
__device__ __forceinline__ void pnp_hyp_body(const float* __restrict__ obj, const float* __restrict__ img,
                                             const int* __restrict__ samples, const double* __restrict__ K,
                                             double* __restrict__ models, int m, float thr, uint8_t* __restrict__ masks,
                                             int* __restrict__ counts, unsigned long long* stamps, const int h) {
    __shared__ EpnpShared sh;
    __shared__ double sR[9], sT[3];
    const int lane = threadIdx.x;
    unsigned long long t_prev = __builtin_readcyclecounter();
#define HSTAMP(k) do { if (stamps && h == 0 && lane == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); stamps[k] += t_ - t_prev; t_prev = t_; } } while (0)
    const double fu = K[0], fv = K[4], uc = K[2], vc = K[5];
    const int n = 5;
    if (lane < n) {   // the five sample points, one lane each (two dependent global round trips instead of ten)
        const double ifx = 1. / fu, ify = 1. / fv;
        const int i = lane;
        const int s = samples[h * 5 + i];
        sh.pws[3 * i] = obj[3 * s]; sh.pws[3 * i + 1] = obj[3 * s + 1]; sh.pws[3 * i + 2] = obj[3 * s + 2];
        const float xn = (float)(((double)img[2 * s] - uc) * ifx);
        const float yn = (float)(((double)img[2 * s + 1] - vc) * ify);
        sh.us[2 * i] = xn * fu + uc;
        sh.us[2 * i + 1] = yn * fv + vc;
    }
    WAVE_SYNC_PNP();
    if (lane == 0) {
        // choose_control_points
        double* cws = sh.cws;
        cws[0] = cws[1] = cws[2] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) cws[j] += sh.pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[j] /= n;
        double C[9], w[3], V[9];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double acc = 0;
                for (int i = 0; i < n; i++) acc += (sh.pws[3 * i + a] - cws[a]) * (sh.pws[3 * i + b] - cws[b]);
                C[a * 3 + b] = acc;
            }
        d_jacobi_eig<3>(C, w, V);
        for (int i = 1; i < 4; i++) {
            const int src = 3 - i;
            const double k = sqrt((w[src] > 0 ? w[src] : 0.0) / n);
            for (int j = 0; j < 3; j++) cws[3 * i + j] = cws[j] + k * V[j * 3 + src];
        }
        // compute_barycentric_coordinates (cvInvert CV_SVD = pseudo-inverse)
        double cc[9], U[9], s3[3], V3[9], ci[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[3 * j + i] - cws[i];
        d_svd3(cc, U, s3, V3);
        const double thr = 2 * DBL_EPSILON * (s3[0] + s3[1] + s3[2]);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int k = 0; k < 3; k++)
                    if (s3[k] > thr) acc += V3[i * 3 + k] * U[j * 3 + k] / s3[k];
                ci[i * 3 + j] = acc;
            }
        for (int i = 0; i < n; i++) {
            const double* pi = &sh.pws[3 * i];
            double* a = &sh.alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0]) + ci[3 * j + 1] * (pi[1] - cws[1]) + ci[3 * j + 2] * (pi[2] - cws[2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
        // fill_M
        for (int i = 0; i < n; i++) {
            const double* as = &sh.alphas[4 * i];
            double* M1 = &sh.M[(2 * i) * 12];
            double* M2 = M1 + 12;
            const double u = sh.us[2 * i], v = sh.us[2 * i + 1];
            for (int k = 0; k < 4; k++) {
                M1[3 * k] = as[k] * fu; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (uc - u);
                M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * fv; M2[3 * k + 2] = as[k] * (vc - v);
            }
        }
    }
    __syncthreads();
    HSTAMP(22);
    // M^T M (upper computed, mirrored) and V = I, one element per lane-step
    for (int idx = lane; idx < 144; idx += 64) {
        const int a = idx / 12, b = idx - a * 12;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        double acc = 0;
        for (int r = 0; r < 2 * n; r++) acc += sh.M[r * 12 + lo] * sh.M[r * 12 + hi];
        sh.A[idx] = acc;
        sh.V[idx] = (a == b) ? 1.0 : 0.0;
    }
    __syncthreads();
    // parallel-order (round-robin) Jacobi: 6 disjoint rotations per round. The block is one wavefront, so LDS ordering
    // (s_waitcnt) is all the synchronisation needed. A <- J^T A J is applied in one pass: the item (g, g') owns the 2x2
    // block {p,q} x {p',q'} (pairs partition the indices, so items touch disjoint elements and update in place), computing
    // first the column rotation of pair g' and then the row rotation of pair g — the same expressions, in the same order,
    // as B = A J followed by A = J^T B. A round in which no pair rotates is skipped (it would multiply by the identity).
    double tol_abs = 0;
    for (int i = 0; i < 12; i++) tol_abs += fabs(sh.A[i * 13]);
    tol_abs *= 1e-17;
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
        for (int r = 0; r < 11; r++) {
            bool rot = false;
            if (lane < 6) {
                const int g = lane;
                int a, b;
                if (g == 0) { a = 11; b = r; }
                else { a = (r + g) % 11; b = (r - g + 11) % 11; }
                const int p = a < b ? a : b, q = a < b ? b : a;
                sh.P[g] = p; sh.Q[g] = q;
                const double apq = sh.A[p * 12 + q], app = sh.A[p * 12 + p], aqq = sh.A[q * 12 + q];
                if (fabs(apq) <= tol_abs) { sh.C[g] = 1.0; sh.S[g] = 0.0; }
                else {
                    const double theta = (aqq - app) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0);
                    sh.C[g] = c;
                    sh.S[g] = t * c;
                    rot = true;
                }
            }
            const int nrot = __popcll(__ballot(rot));
            rotated += nrot;
            WAVE_SYNC_PNP();
            if (nrot == 0) continue;   // wave-uniform
            if (lane < 36) {
                const int g = lane / 6, g2 = lane - g * 6;
                const int p = sh.P[g], q = sh.Q[g], p2 = sh.P[g2], q2 = sh.Q[g2];
                const double c = sh.C[g], sn = sh.S[g], c2 = sh.C[g2], sn2 = sh.S[g2];
                const double app = sh.A[p * 12 + p2], apq = sh.A[p * 12 + q2], aqp = sh.A[q * 12 + p2], aqq = sh.A[q * 12 + q2];
                const double bpp = c2 * app - sn2 * apq, bpq = sn2 * app + c2 * apq;   // B = A J, rows p and q
                const double bqp = c2 * aqp - sn2 * aqq, bqq = sn2 * aqp + c2 * aqq;
                sh.A[p * 12 + p2] = c * bpp - sn * bqp;                                 // A = J^T B
                sh.A[q * 12 + p2] = sn * bpp + c * bqp;
                sh.A[p * 12 + q2] = c * bpq - sn * bqq;
                sh.A[q * 12 + q2] = sn * bpq + c * bqq;
            }
            for (int idx = lane; idx < 72; idx += 64) {   // V = V J, item = (row i, pair g)
                const int i = idx / 6, g = idx - i * 6;
                const int p = sh.P[g], q = sh.Q[g];
                const double c = sh.C[g], sn = sh.S[g];
                const double x = sh.V[i * 12 + p], y = sh.V[i * 12 + q];
                sh.V[i * 12 + p] = c * x - sn * y;
                sh.V[i * 12 + q] = sn * x + c * y;
            }
            WAVE_SYNC_PNP();
        }
        if (!rotated) break;
    }
    HSTAMP(23);
    if (lane == 0) {
        // ascending selection sort on the diagonal, keep the 4 smallest eigenvectors
        double w[12];
        int ord[12];
        for (int i = 0; i < 12; i++) { w[i] = sh.A[i * 12 + i]; ord[i] = i; }
        for (int i = 0; i < 11; i++) {
            int m = i;
            for (int j = i + 1; j < 12; j++) if (w[j] < w[m]) m = j;
            if (m != i) { double t = w[i]; w[i] = w[m]; w[m] = t; int o = ord[i]; ord[i] = ord[m]; ord[m] = o; }
        }
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 12; k++) sh.v4[i * 12 + k] = sh.V[k * 12 + ord[i]];
        const double* cws = sh.cws;   // compute_rho
        double* rho = sh.rho;
        rho[0] = d_dist2(cws, cws + 3); rho[1] = d_dist2(cws, cws + 6); rho[2] = d_dist2(cws, cws + 9);
        rho[3] = d_dist2(cws + 3, cws + 6); rho[4] = d_dist2(cws + 3, cws + 9); rho[5] = d_dist2(cws + 6, cws + 9);
    }
    WAVE_SYNC_PNP();
    if (lane < 60) {
        // compute_L_6x10, one entry per lane: row i = control-point pair (a,b), column k = product of the difference vectors of
        // null-space vectors (u,w) in epnp's order 00 01 11 02 12 22 03 13 23 33 (off-diagonal ones doubled)
        const int i = lane / 10, k = lane - i * 10;
        const int pa = (i < 3) ? 0 : ((i < 5) ? 1 : 2), pb = (i < 3) ? i + 1 : ((i < 5) ? i - 1 : 3);
        const int u = (k == 0 || k == 1 || k == 3 || k == 6) ? 0 : ((k == 2 || k == 4 || k == 7) ? 1 : ((k == 5 || k == 8) ? 2 : 3));
        const int w = (k == 0) ? 0 : ((k == 1 || k == 2) ? 1 : ((k == 3 || k == 4 || k == 5) ? 2 : 3));
        const double* vu = sh.v4 + 12 * u;
        const double* vw = sh.v4 + 12 * w;
        const double du[3] = {vu[3 * pa] - vu[3 * pb], vu[3 * pa + 1] - vu[3 * pb + 1], vu[3 * pa + 2] - vu[3 * pb + 2]};
        const double dw[3] = {vw[3 * pa] - vw[3 * pb], vw[3 * pa + 1] - vw[3 * pb + 1], vw[3 * pa + 2] - vw[3 * pb + 2]};
        const double d = d_dot3(du, dw);
        sh.L[i * 10 + k] = (u == w) ? d : 2.0f * d;
    }
    __syncthreads();
    HSTAMP(24);
    // the three EPnP cases (N = 1, 2, 3 null-space vectors) are independent: lanes 0..2 evaluate them side by side
    if (lane < 3) {
        const int N = lane + 1;
        double Lr[LDS_L ? 1 : 60], rhor[LDS_L ? 1 : 6], betas[4];
        if (!LDS_L) {
            for (int i = 0; i < 60; i++) Lr[i] = sh.L[i];
            for (int i = 0; i < 6; i++) rhor[i] = sh.rho[i];
        }
        const double* L = LDS_L ? sh.L : Lr;
        const double* rho = LDS_L ? sh.rho : rhor;
        // find_betas_approx_1/2/3: least squares on 4 / 3 / 5 columns of L_6x10. The three lanes run ONE code path (lanes of a
        // wavefront that take different paths execute them one after the other): the 6x4 and 6x3 systems are padded with zero
        // columns to 6x5. Padding is exact: the padded rows/columns of A^T A are zero and stay zero, the cyclic Jacobi skips
        // their pairs (apq == 0), which leaves exactly the rotation sequence of the smaller matrix; the padded eigenvalue 0 is
        // below the singular-value threshold and contributes nothing.
        double l5[30], b5[5];
        {
            const int c0 = 0, c1 = 1, c2 = (N == 1) ? 3 : 2, c3 = (N == 1) ? 6 : ((N == 3) ? 3 : -1), c4 = (N == 3) ? 4 : -1;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                l5[i * 5 + 0] = sh.L[i * 10 + c0];
                l5[i * 5 + 1] = sh.L[i * 10 + c1];
                l5[i * 5 + 2] = sh.L[i * 10 + c2];
                l5[i * 5 + 3] = (c3 >= 0) ? sh.L[i * 10 + (c3 >= 0 ? c3 : 0)] : 0.0;
                l5[i * 5 + 4] = (c4 >= 0) ? sh.L[i * 10 + (c4 >= 0 ? c4 : 0)] : 0.0;
            }
        }
        d_pinv_solve<6, 5>(l5, rho, b5);
        if (N == 1) {
            if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = -b5[1] / betas[0]; betas[2] = -b5[2] / betas[0]; betas[3] = -b5[3] / betas[0]; }
            else { betas[0] = sqrt(b5[0]); betas[1] = b5[1] / betas[0]; betas[2] = b5[2] / betas[0]; betas[3] = b5[3] / betas[0]; }
        } else if (N == 2) {
            if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
            else { betas[0] = sqrt(b5[0]); betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) betas[0] = -betas[0];
            betas[2] = 0.0; betas[3] = 0.0;
        } else {
            if (b5[0] < 0) { betas[0] = sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0; }
            else { betas[0] = sqrt(b5[0]); betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0; }
            if (b5[1] < 0) betas[0] = -betas[0];
            betas[2] = b5[3] / betas[0];
            betas[3] = 0.0;
        }
        HSTAMP(25);
        epnp_gauss_newton(L, rho, betas);
        HSTAMP(26);
        double R[9], t[3];
        sh.case_err[lane] = epnp_R_and_t(sh, betas, fu, fv, uc, vc, R, t);
        for (int i = 0; i < 9; i++) sh.case_R[lane * 9 + i] = R[i];
        for (int i = 0; i < 3; i++) sh.case_t[lane * 3 + i] = t[i];
    }
    __syncthreads();
    HSTAMP(27);
    if (lane == 0) {
        // N = 1; if (rep[2] < rep[1]) N = 2; if (rep[3] < rep[N]) N = 3;
        int N = 0;
        if (sh.case_err[1] < sh.case_err[0]) N = 1;
        if (sh.case_err[2] < sh.case_err[N]) N = 2;
        double rv[3];
        d_rodrigues_m2v(sh.case_R + 9 * N, rv);
        for (int i = 0; i < 3; i++) { models[h * 6 + i] = rv[i]; models[h * 6 + 3 + i] = sh.case_t[N * 3 + i]; sT[i] = sh.case_t[N * 3 + i]; }
        d_rodrigues_v2m(rv, sR);   // the score uses the rotation of the stored vector (PnPRansacCallback::computeError: projectPoints(rvec))
    }
    __syncthreads();
    HSTAMP(28);
    // float32 squared reprojection error of every point under this hypothesis, inlier mask and count
    {
        const double t0 = sT[0], t1 = sT[1], t2 = sT[2];
        int good = 0;
        for (int i = lane; i < m; i += 64) {
            const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
            double x = sR[0] * X + sR[1] * Y + sR[2] * Z + t0;
            double y = sR[3] * X + sR[4] * Y + sR[5] * Z + t1;
            double z = sR[6] * X + sR[7] * Y + sR[8] * Z + t2;
            z = z ? 1. / z : 1;
            x *= z; y *= z;
            const float px = (float)(x * fu + uc), py = (float)(y * fv + vc);
            const float dx = img[2 * i] - px, dy = img[2 * i + 1] - py;
            const float e = dx * dx + dy * dy;
            const int f = e <= thr;
            masks[(size_t)h * m + i] = (uint8_t)f;
            good += f;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) good += __shfl_xor(good, o, 64);
        if (lane == 0) counts[h] = good;
    }
    HSTAMP(29);
#undef HSTAMP
}

// ---- RANSAC bookkeeping + LM refit (one 256-thread workgroup) -----------------------------------------------------------
__device__ inline int d_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

__device__ inline void d_angle_axis_rotate_jac(const double a[3], const double q[3], double p[3], double dpdw[9]) {
    const double theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        const double ct = cos(theta), st = sin(theta), ti = 1.0 / theta;
        const double w[3] = {a[0] * ti, a[1] * ti, a[2] * ti};
        const double wxq[3] = {w[1] * q[2] - w[2] * q[1], w[2] * q[0] - w[0] * q[2], w[0] * q[1] - w[1] * q[0]};
        const double wq = w[0] * q[0] + w[1] * q[1] + w[2] * q[2];
        const double tmp = wq * (1.0 - ct);
        for (int i = 0; i < 3; i++) p[i] = q[i] * ct + wxq[i] * st + w[i] * tmp;
        if (!dpdw) return;
        for (int k = 0; k < 3; k++) {
            double dwk[3];
            for (int i = 0; i < 3; i++) dwk[i] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) * ti;
            const double dwxq[3] = {dwk[1] * q[2] - dwk[2] * q[1], dwk[2] * q[0] - dwk[0] * q[2], dwk[0] * q[1] - dwk[1] * q[0]};
            const double dwq = dwk[0] * q[0] + dwk[1] * q[1] + dwk[2] * q[2];
            const double dct = -st * w[k], dst = ct * w[k];
            const double dtmp = dwq * (1.0 - ct) + wq * (st * w[k]);
            for (int i = 0; i < 3; i++) dpdw[i * 3 + k] = q[i] * dct + dwxq[i] * st + wxq[i] * dst + dwk[i] * tmp + w[i] * dtmp;
        }
    } else {
        const double wxq[3] = {a[1] * q[2] - a[2] * q[1], a[2] * q[0] - a[0] * q[2], a[0] * q[1] - a[1] * q[0]};
        for (int i = 0; i < 3; i++) p[i] = q[i] + wxq[i];
        if (!dpdw) return;
        dpdw[0] = 0;     dpdw[1] = q[2];  dpdw[2] = -q[1];
        dpdw[3] = -q[2]; dpdw[4] = 0;     dpdw[5] = q[0];
        dpdw[6] = q[1];  dpdw[7] = -q[0]; dpdw[8] = 0;
    }
}

constexpr int RF_T = 256;
struct RefitShared {
    double tile[28 * (256 + 8)];   // per-thread partials, [value][thread], rows padded by 8 (bank spread)
    double part[28 * 8];
    double red[4 * 28];
    double JtJ[36], JtErr[6], param[6], prev[6];
    double err2, prevErr2;
    int best, last, n_in, state, lambdaLg10, iters, done;
    int wsum[4];
};

// sums 28 per-thread values over the block in a fixed order: LDS tile [28][256 (+8 pad)], 28 x 8 threads add 32 entries each
// (entry j*8 + part: consecutive lanes read consecutive addresses, the 8-double row pad spreads the 8 values k of a wavefront
// over different banks — a contiguous 32-entry chunk per lane put all 64 lanes on one bank), 28 threads add the 8 partials.
// Result in sh.red[0..27].
constexpr int RF_ROW = RF_T + 8;
__device__ inline void block_sum28(const double* acc, RefitShared& sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 28; k++) sh.tile[k * RF_ROW + tid] = acc[k];
    __syncthreads();
    if (tid < 28 * 8) {
        const int k = tid >> 3, part = tid & 7;
        const double* src = &sh.tile[k * RF_ROW + part];
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;   // four independent chains (fixed order), 8 dependent adds instead of 32
#pragma unroll
        for (int j = 0; j < 32; j += 4) { s0 += src[8 * j]; s1 += src[8 * (j + 1)]; s2 += src[8 * (j + 2)]; s3 += src[8 * (j + 3)]; }
        sh.part[k * 8 + part] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < 28) {
        double sacc = sh.part[tid * 8];
#pragma unroll
        for (int q = 1; q < 8; q++) sacc += sh.part[tid * 8 + q];
        sh.red[tid] = sacc;
    }
    __syncthreads();
}

__device__ __forceinline__ void pnp_select_refit_body(const float* __restrict__ obj, const float* __restrict__ img, int m,
                                                      const double* __restrict__ K, const double* __restrict__ models,
                                                      const uint8_t* __restrict__ masks, const int* __restrict__ counts,
                                                      int n_hyp, double confidence, double* __restrict__ rt_out,
                                                      int* __restrict__ inliers, int* __restrict__ info, char* __restrict__ host_out,
                                                      unsigned long long* stamps, const unsigned done_seq) {
    // done_seq != 0: after the result block in mapped pinned memory is complete, its last info word takes this call's sequence number -
    // the host spins on that word instead of asking the runtime to synchronise the stream (a marker packet and its signal later)
    __shared__ RefitShared sh;
    const unsigned long long t_start = __builtin_readcyclecounter();
    unsigned long long t_lm0 = 0;
    int n_pass = 0;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        // RANSACPointSetRegistrator::run replayed over the precomputed hypotheses
        int maxGood = 0, niters = n_hyp > 1 ? n_hyp : 1, best = -1, last = -1;
        for (int iter = 0; iter < niters && iter < n_hyp; iter++) {
            last = iter;
            const int good = counts[iter];
            if (good > (maxGood > 4 ? maxGood : 4)) {
                best = iter;
                maxGood = good;
                niters = d_ransac_update_num_iters(confidence, (double)(m - good) / m, 5, niters);
            }
        }
        sh.best = best; sh.last = last; sh.n_in = 0;
        info[1] = last + 1;   // hypotheses the sequential algorithm would have evaluated
    }
    __syncthreads();
    if (sh.best < 0) {
        if (tid == 0) {
            info[0] = 0;
            const int l = sh.last < 0 ? 0 : sh.last;
            for (int i = 0; i < 6; i++) rt_out[i] = models[l * 6 + i];
            if (host_out) {
                for (int i = 0; i < 6; i++) ((double*)host_out)[i] = models[l * 6 + i];
                ((int*)(host_out + 48))[0] = 0; ((int*)(host_out + 48))[1] = sh.last + 1;
                if (done_seq) { __threadfence_system(); __hip_atomic_store((unsigned*)(host_out + 60), done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
            }
        }
        return;
    }
    // ordered compaction of the best mask
    const uint8_t* mk = masks + (size_t)sh.best * m;
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += RF_T) {
        const int i = c0 + tid;
        const int f = (i < m) ? mk[i] : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) sh.wsum[wid] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wid; w++) off += sh.wsum[w];
        if (f) inliers[off + before] = i;
        const int tot = sh.wsum[0] + sh.wsum[1] + sh.wsum[2] + sh.wsum[3];
        __syncthreads();
        base += tot;
    }
    const int n = base;
    if (tid == 0) { sh.n_in = n; info[0] = n; }
    // ---- cvFindExtrinsicCameraParams2 (useExtrinsicGuess): CvLevMarq state machine, initial guess = last evaluated model
    if (tid < 6) { sh.param[tid] = models[sh.last * 6 + tid]; }
    if (tid == 0) { sh.lambdaLg10 = -3; sh.iters = 0; sh.done = 0; sh.state = 0; }
    __syncthreads();
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    // this thread's first two inliers stay in registers for all LM passes (n <= 512 is the rule: no pass waits for the index -> point
    // chain of global loads again); further ones are re-read per pass. Same values, same accumulation order.
    constexpr int RF_KEEP = 2;
    double keepX[RF_KEEP][3], keepU[RF_KEEP][2];
#pragma unroll
    for (int q = 0; q < RF_KEEP; q++) {
        const int e = tid + q * RF_T;
        const int i = e < n ? inliers[e] : 0;
        keepX[q][0] = (double)obj[3 * i]; keepX[q][1] = (double)obj[3 * i + 1]; keepX[q][2] = (double)obj[3 * i + 2];
        keepU[q][0] = (double)img[2 * i]; keepU[q][1] = (double)img[2 * i + 1];
    }
    // state 0: compute J & err at param, step; state 1: check err at new param
    t_lm0 = __builtin_readcyclecounter();
    for (;;) {
        n_pass++;
        unsigned long long t_p = __builtin_readcyclecounter();
#define RSTAMP(k) do { if (stamps && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); stamps[k] += t_ - t_p; t_p = t_; } } while (0)
        const bool jac = (sh.state == 0);
        double acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0;
        const double p0 = sh.param[0], p1 = sh.param[1], p2 = sh.param[2], p3 = sh.param[3], p4 = sh.param[4], p5 = sh.param[5];
        const double pa[3] = {p0, p1, p2};
        for (int e = tid, q = 0; e < n; e += RF_T, q++) {
            double X[3], U[2];
            if (q == 0) { X[0] = keepX[0][0]; X[1] = keepX[0][1]; X[2] = keepX[0][2]; U[0] = keepU[0][0]; U[1] = keepU[0][1]; }
            else if (q == 1) { X[0] = keepX[1][0]; X[1] = keepX[1][1]; X[2] = keepX[1][2]; U[0] = keepU[1][0]; U[1] = keepU[1][1]; }
            else {
                const int i = inliers[e];
                X[0] = (double)obj[3 * i]; X[1] = (double)obj[3 * i + 1]; X[2] = (double)obj[3 * i + 2];
                U[0] = (double)img[2 * i]; U[1] = (double)img[2 * i + 1];
            }
            double Xc[3], dpdw[9];
            d_angle_axis_rotate_jac(pa, X, Xc, jac ? dpdw : nullptr);
            Xc[0] += p3; Xc[1] += p4; Xc[2] += p5;
            const double z = Xc[2] ? 1. / Xc[2] : 1;
            const double x = Xc[0] * z, y = Xc[1] * z;
            const double ex = x * fx + cx - U[0], ey = y * fy + cy - U[1];
            acc[27] += ex * ex + ey * ey;
            if (jac) {
                double Ju[6], Jv[6];
                const double du[3] = {fx * z, 0, -fx * x * z}, dv[3] = {0, fy * z, -fy * y * z};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    Ju[k] = du[0] * dpdw[k] + du[1] * dpdw[3 + k] + du[2] * dpdw[6 + k];
                    Jv[k] = dv[0] * dpdw[k] + dv[1] * dpdw[3 + k] + dv[2] * dpdw[6 + k];
                    Ju[3 + k] = du[k];
                    Jv[3 + k] = dv[k];
                }
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int b = a; b < 6; b++) acc[k++] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
                }
#pragma unroll
                for (int a = 0; a < 6; a++) acc[21 + a] += Ju[a] * ex + Jv[a] * ey;
            }
        }
        RSTAMP(10);
        block_sum28(acc, sh);
        RSTAMP(11);
        if (tid == 0) {
            const double err2 = sh.red[27];
            bool do_step = false;
            if (sh.state == 0) {
                int k = 0;
                for (int a = 0; a < 6; a++)
                    for (int b = a; b < 6; b++) { sh.JtJ[a * 6 + b] = sh.JtJ[b * 6 + a] = sh.red[k]; k++; }
                for (int a = 0; a < 6; a++) { sh.JtErr[a] = sh.red[21 + a]; sh.prev[a] = sh.param[a]; }
                if (sh.iters == 0) sh.prevErr2 = err2;
                do_step = true;
                sh.state = 1;
            } else {
                bool retry = false;
                if (err2 > sh.prevErr2) {
                    if (++sh.lambdaLg10 <= 16) { do_step = true; retry = true; }
                }
                if (!retry) {
                    sh.lambdaLg10 = sh.lambdaLg10 - 1 > -16 ? sh.lambdaLg10 - 1 : -16;
                    double dn = 0, pn = 0;
                    for (int i = 0; i < 6; i++) { dn += (sh.param[i] - sh.prev[i]) * (sh.param[i] - sh.prev[i]); pn += sh.prev[i] * sh.prev[i]; }
                    if (++sh.iters >= 20 || sqrt(dn) / sqrt(pn) < (double)FLT_EPSILON) sh.done = 1;
                    else { sh.prevErr2 = err2; sh.state = 0; }
                }
            }
            if (do_step) {
                // (JtJ + lambda diag(JtJ)) x = JtErr by Gaussian elimination with partial pivoting on [A | b], entirely in this
                // thread's registers (every index is a compile-time constant after unrolling): the same element-wise expressions
                // as the row-by-row algorithm, without the LDS round trips of a lane-per-row version.
                const double lambda = K[10 + 16 + sh.lambdaLg10];   // exp(lambdaLg10 * log(10.)) tabulated by the host (|lambdaLg10| <= 16)
                double a[6][7];
#pragma unroll
                for (int i = 0; i < 6; i++) {
#pragma unroll
                    for (int k = 0; k < 6; k++) a[i][k] = sh.JtJ[i * 6 + k];
                    a[i][i] *= 1. + lambda;
                    a[i][6] = sh.JtErr[i];
                }
                bool ok = true;
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    if (!ok) continue;
                    int piv = c;
                    double best = fabs(a[c][c]);
#pragma unroll
                    for (int q = c + 1; q < 6; q++) { const double v = fabs(a[q][c]); if (v > best) { best = v; piv = q; } }
                    if (best == 0.0) { ok = false; continue; }
#pragma unroll
                    for (int q = c + 1; q < 6; q++) {
                        if (piv == q) {
#pragma unroll
                            for (int k = 0; k < 7; k++) { const double t = a[c][k]; a[c][k] = a[q][k]; a[q][k] = t; }
                        }
                    }
#pragma unroll
                    for (int r = c + 1; r < 6; r++) {
                        const double f = a[r][c] / a[c][c];
                        if (f != 0.0) {
#pragma unroll
                            for (int k = c; k < 7; k++) a[r][k] -= f * a[c][k];
                        }
                    }
                }
                double bb[6] = {0, 0, 0, 0, 0, 0};
                if (ok) {
#pragma unroll
                    for (int q = 5; q >= 0; q--) {
                        double v = a[q][6];
#pragma unroll
                        for (int k = q + 1; k < 6; k++) v -= a[q][k] * bb[k];
                        bb[q] = v / a[q][q];
                    }
                }
#pragma unroll
                for (int i = 0; i < 6; i++) sh.param[i] = sh.prev[i] - bb[i];
            }
        }
        __syncthreads();
        RSTAMP(12);
        if (sh.done) break;
    }
#undef RSTAMP
    if (stamps && tid == 0) {   // diagnostic: [30] cycles before the LM loop, [31] cycles in it, [21] passes
        const unsigned long long t_end = __builtin_readcyclecounter();
        stamps[30] += t_lm0 - t_start; stamps[31] += t_end - t_lm0; stamps[21] += n_pass;
    }
    if (tid < 6) rt_out[tid] = sh.param[tid];
    if (host_out) {   // the result block in mapped pinned host memory: no device-to-host copy afterwards
        if (tid < 6) ((double*)host_out)[tid] = sh.param[tid];
        if (tid == 0) { ((int*)(host_out + 48))[0] = n; ((int*)(host_out + 48))[1] = sh.last + 1; }
        int* hin = (int*)(host_out + 64);
        for (int e = tid; e < n; e += RF_T) hin[e] = inliers[e];
        if (done_seq) {
            __threadfence_system();
            __syncthreads();
            if (tid == 0) __hip_atomic_store((unsigned*)(host_out + 60), done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// grid = n_hyp, block = 64
__global__ __launch_bounds__(64) void k_pnp_hyp(const float* __restrict__ obj, const float* __restrict__ img,
                                                const int* __restrict__ samples, const double* __restrict__ K,
                                                double* __restrict__ models, int m, float thr, uint8_t* __restrict__ masks,
                                                int* __restrict__ counts, unsigned long long* stamps) { BACKEND_PRIO();
    pnp_hyp_body<false>(obj, img, samples, K, models, m, thr, masks, counts, stamps, blockIdx.x);
}
__global__ __launch_bounds__(RF_T) void k_pnp_select_refit(const float* __restrict__ obj, const float* __restrict__ img, int m,
                                                           const double* __restrict__ K, const double* __restrict__ models,
                                                           const uint8_t* __restrict__ masks, const int* __restrict__ counts,
                                                           int n_hyp, double confidence, double* __restrict__ rt_out,
                                                           int* __restrict__ inliers, int* __restrict__ info, char* __restrict__ host_out,
                                                           unsigned long long* stamps, unsigned done_seq) { BACKEND_PRIO();
    pnp_select_refit_body(obj, img, m, K, models, masks, counts, n_hyp, confidence, rt_out, inliers, info, host_out, stamps, done_seq);
}
// batched forms: blockIdx.y = problem (several sequences' PnP calls in one launch), same per-problem arithmetic
// Two launches per round: hypotheses [0, h0) and [h0, n_hyp). A wavefront of the second launch first replays the sequential RANSAC
// bookkeeping over the counts of the first h0 hypotheses (the same loop as pnp_select_refit_body): when the adaptive iteration count
// has already dropped to <= h0, the sequential algorithm never reaches this hypothesis, nothing downstream reads its model, mask or
// count, and the wavefront returns at once. (cv::solvePnPRansac on typical inlier ratios stops after ~10-20 of its 100 iterations;
// the single-sequence launch evaluates all 100 side by side because it is bound by one hypothesis' latency, the batched launch
// is bound by how many wavefronts the chip can hold.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pnp_hyp_batch(const PnPProblem* __restrict__ probs, int h0, int replay) { BACKEND_PRIO();
    const PnPProblem p = probs[blockIdx.y];
    const int h = h0 + (int)blockIdx.x;
    if (h >= p.n_hyp) return;
    if (replay) {
        int maxGood = 0, niters = p.n_hyp > 1 ? p.n_hyp : 1;
        for (int iter = 0; iter < niters && iter < h0; iter++) {
            const int good = p.counts[iter];
            if (good > (maxGood > 4 ? maxGood : 4)) {
                maxGood = good;
                niters = d_ransac_update_num_iters(p.confidence, (double)(p.m - good) / p.m, 5, niters);
            }
        }
        if (niters <= h0) return;   // wave-uniform
    }
    pnp_hyp_body<true>(p.obj, p.img, p.samples, p.K, p.models, p.m, p.thr, p.masks, p.counts, nullptr, h);
}
__global__ __launch_bounds__(RF_T) void k_pnp_select_refit_batch(const PnPProblem* __restrict__ probs) { BACKEND_PRIO();
    const PnPProblem p = probs[blockIdx.x];
    pnp_select_refit_body(p.obj, p.img, p.m, p.K, p.models, p.masks, p.counts, p.n_hyp, p.confidence, p.rt_out, p.inliers, p.info, p.host_out, nullptr, 0u);
}

hipError_t launch_pnp_batch(hipStream_t s, const PnPProblem* d_probs, int n_probs, int max_hyp) {
    if (n_probs <= 0) return hipSuccess;
    if (!d_probs || max_hyp < 1) return hipErrorInvalidValue;
    static const int stage1 = getenv("PMV_PNP_STAGE1") ? atoi(getenv("PMV_PNP_STAGE1")) : 16;
    const int h0 = stage1 > 0 && stage1 < max_hyp ? stage1 : max_hyp;
    { ProfScope ps(K_PNP_HYP, s);
    hipLaunchKernelGGL(k_pnp_hyp_batch, dim3(h0, n_probs), dim3(64), 0, s, d_probs, 0, 0); }
    if (h0 < max_hyp) {
        ProfScope ps(K_PNP_HYP, s);
        hipLaunchKernelGGL(k_pnp_hyp_batch, dim3(max_hyp - h0, n_probs), dim3(64), 0, s, d_probs, h0, 1);
    }
    ProfScope ps3(K_PNP_REFIT, s);
    hipLaunchKernelGGL(k_pnp_select_refit_batch, dim3(n_probs), dim3(RF_T), 0, s, d_probs);
    return hipGetLastError();
}

hipError_t launch_pnp(hipStream_t s, const float* d_obj, const float* d_img, int m, const double* d_K, const int* d_samples,
                      int n_hyp, float thr, double confidence, double* d_models, uint8_t* d_masks, int* d_counts,
                      double* d_rt_out, int* d_inliers, int* d_info, char* host_out, unsigned long long* d_stamps, unsigned done_seq) {
    { ProfScope ps(K_PNP_HYP, s);
    hipLaunchKernelGGL(k_pnp_hyp, dim3(n_hyp), dim3(64), 0, s, d_obj, d_img, d_samples, d_K, d_models, m, thr, d_masks, d_counts, d_stamps); }
    ProfScope ps3(K_PNP_REFIT, s);
    hipLaunchKernelGGL(k_pnp_select_refit, dim3(1), dim3(RF_T), 0, s, d_obj, d_img, m, d_K, d_models, d_masks, d_counts, n_hyp,
                       confidence, d_rt_out, d_inliers, d_info, host_out, d_stamps, done_seq);
    return hipGetLastError();
}

}  // namespace pmv
