// gfx950 bundle adjustment: ProjectionResidual + analytic Jacobians (ProjectionResidual.h:38-58) and the whole
// Levenberg–Marquardt solve that the reference delegates to ceres::Solve (CeresBundleAdjustment.cpp:50-61), resident on the
// device: ONE launch runs every LM iteration (no host round trip per iteration).
//
// Structure of one iteration (one 512-thread workgroup; all cross-thread sums use a fixed tree => run-to-run bitwise
// reproducible):
//   evaluate r, J (Huber corrector)  ->  Jacobi column scaling  ->  LM diagonal  ->  per-camera blocks U_c, rhs_c  ->
//   per-point blocks: E_p^-1, g_p, W_p = Jc^T Jp, Y_p = W_p E_p^-1 written as dense rows of Yd / [Wd | g]  ->
//   Schur complement  S -= Yd^T Wd,  rhs -= Yd^T g   as ONE dense contraction on FP64 MFMA (v_mfma_f64_16x16x4_f64)  ->
//   Cholesky of the reduced camera matrix -> back-substitution -> model cost change -> candidate cost -> accept/reject.
#include "pmv_ctx.h"
#include <algorithm>
#include "backend.h"
#include "pmv_prof.h"
#include <float.h>

namespace pmv {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BA_T = 512;
constexpr int BA_NW = BA_T / 64;


// ---- ceres::AngleAxisRotatePoint + exact derivatives (both branches) -------------------------------------------------
__device__ inline void angle_axis_rotate(const double a[3], const double q[3], double p[3], double dpdw[9], double Rm[9], bool jac) {
    const double theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        const double ct = cos(theta), st = sin(theta), ti = 1.0 / theta;
        const double w[3] = {a[0] * ti, a[1] * ti, a[2] * ti};
        const double wxq[3] = {w[1] * q[2] - w[2] * q[1], w[2] * q[0] - w[0] * q[2], w[0] * q[1] - w[1] * q[0]};
        const double wq = w[0] * q[0] + w[1] * q[1] + w[2] * q[2];
        const double tmp = wq * (1.0 - ct);
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] = q[i] * ct + wxq[i] * st + w[i] * tmp;
        if (!jac) return;
        const double c1 = 1.0 - ct;
        Rm[0] = ct + c1 * w[0] * w[0];        Rm[1] = c1 * w[0] * w[1] - st * w[2]; Rm[2] = c1 * w[0] * w[2] + st * w[1];
        Rm[3] = c1 * w[1] * w[0] + st * w[2]; Rm[4] = ct + c1 * w[1] * w[1];        Rm[5] = c1 * w[1] * w[2] - st * w[0];
        Rm[6] = c1 * w[2] * w[0] - st * w[1]; Rm[7] = c1 * w[2] * w[1] + st * w[0]; Rm[8] = ct + c1 * w[2] * w[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double dwk[3];
#pragma unroll
            for (int i = 0; i < 3; i++) dwk[i] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) * ti;
            const double dwxq[3] = {dwk[1] * q[2] - dwk[2] * q[1], dwk[2] * q[0] - dwk[0] * q[2], dwk[0] * q[1] - dwk[1] * q[0]};
            const double dwq = dwk[0] * q[0] + dwk[1] * q[1] + dwk[2] * q[2];
            const double dct = -st * w[k], dst = ct * w[k];
            const double dtmp = dwq * (1.0 - ct) + wq * (st * w[k]);
#pragma unroll
            for (int i = 0; i < 3; i++) dpdw[i * 3 + k] = q[i] * dct + dwxq[i] * st + wxq[i] * dst + dwk[i] * tmp + w[i] * dtmp;
        }
    } else {
        const double wxq[3] = {a[1] * q[2] - a[2] * q[1], a[2] * q[0] - a[0] * q[2], a[0] * q[1] - a[1] * q[0]};
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] = q[i] + wxq[i];
        if (!jac) return;
        dpdw[0] = 0;     dpdw[1] = q[2];  dpdw[2] = -q[1];
        dpdw[3] = -q[2]; dpdw[4] = 0;     dpdw[5] = q[0];
        dpdw[6] = q[1];  dpdw[7] = -q[0]; dpdw[8] = 0;
        Rm[0] = 1;     Rm[1] = -a[2]; Rm[2] = a[1];
        Rm[3] = a[2];  Rm[4] = 1;     Rm[5] = -a[0];
        Rm[6] = -a[1]; Rm[7] = a[0];  Rm[8] = 1;
    }
}

// Per-camera constants of AngleAxisRotatePoint: theta-dependent terms are identical for every observation of a camera, so
// they are evaluated once per camera and evaluation point (same expressions, same bits as evaluating them per observation).
struct CamRot { double ct, st, ti, w0, w1, w2; int big; int pad; };
__device__ inline void cam_rot_setup(const double a[3], CamRot& c) {
    const double theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        c.ct = cos(theta); c.st = sin(theta); c.ti = 1.0 / theta;
        c.w0 = a[0] * c.ti; c.w1 = a[1] * c.ti; c.w2 = a[2] * c.ti;
        c.big = 1;
    } else { c.ct = 1; c.st = 0; c.ti = 0; c.w0 = a[0]; c.w1 = a[1]; c.w2 = a[2]; c.big = 0; }
}
__device__ inline void angle_axis_rotate_pre(const CamRot& c, const double q[3], double p[3], double dpdw[9], double Rm[9], bool jac) {
    if (c.big) {
        const double ct = c.ct, st = c.st, ti = c.ti;
        const double w[3] = {c.w0, c.w1, c.w2};
        const double wxq[3] = {w[1] * q[2] - w[2] * q[1], w[2] * q[0] - w[0] * q[2], w[0] * q[1] - w[1] * q[0]};
        const double wq = w[0] * q[0] + w[1] * q[1] + w[2] * q[2];
        const double tmp = wq * (1.0 - ct);
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] = q[i] * ct + wxq[i] * st + w[i] * tmp;
        if (!jac) return;
        const double c1 = 1.0 - ct;
        Rm[0] = ct + c1 * w[0] * w[0];        Rm[1] = c1 * w[0] * w[1] - st * w[2]; Rm[2] = c1 * w[0] * w[2] + st * w[1];
        Rm[3] = c1 * w[1] * w[0] + st * w[2]; Rm[4] = ct + c1 * w[1] * w[1];        Rm[5] = c1 * w[1] * w[2] - st * w[0];
        Rm[6] = c1 * w[2] * w[0] - st * w[1]; Rm[7] = c1 * w[2] * w[1] + st * w[0]; Rm[8] = ct + c1 * w[2] * w[2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double dwk[3];
#pragma unroll
            for (int i = 0; i < 3; i++) dwk[i] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) * ti;
            const double dwxq[3] = {dwk[1] * q[2] - dwk[2] * q[1], dwk[2] * q[0] - dwk[0] * q[2], dwk[0] * q[1] - dwk[1] * q[0]};
            const double dwq = dwk[0] * q[0] + dwk[1] * q[1] + dwk[2] * q[2];
            const double dct = -st * w[k], dst = ct * w[k];
            const double dtmp = dwq * (1.0 - ct) + wq * (st * w[k]);
#pragma unroll
            for (int i = 0; i < 3; i++) dpdw[i * 3 + k] = q[i] * dct + dwxq[i] * st + wxq[i] * dst + dwk[i] * tmp + w[i] * dtmp;
        }
    } else {
        const double a[3] = {c.w0, c.w1, c.w2};
        const double wxq[3] = {a[1] * q[2] - a[2] * q[1], a[2] * q[0] - a[0] * q[2], a[0] * q[1] - a[1] * q[0]};
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] = q[i] + wxq[i];
        if (!jac) return;
        dpdw[0] = 0;     dpdw[1] = q[2];  dpdw[2] = -q[1];
        dpdw[3] = -q[2]; dpdw[4] = 0;     dpdw[5] = q[0];
        dpdw[6] = q[1];  dpdw[7] = -q[0]; dpdw[8] = 0;
        Rm[0] = 1;     Rm[1] = -a[2]; Rm[2] = a[1];
        Rm[3] = a[2];  Rm[4] = 1;     Rm[5] = -a[0];
        Rm[6] = -a[1]; Rm[7] = a[0];  Rm[8] = 1;
    }
}
// same as projection_residual with the camera's rotation constants precomputed
__device__ inline void projection_residual_pre(const CamRot& cr, const double* cam, const double* X, double ox, double oy, const double* K,
                                               double r[2], double* Jc, double* Jp, bool jac) {
    const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    const double q[3] = {X[0] + cam[3], X[1] + cam[4], X[2] + cam[5]};
    double p[3], dpdw[9], Rm[9];
    angle_axis_rotate_pre(cr, q, p, dpdw, Rm, jac);
    const double pz = p[2] * -1.0;
    const double u = p[0] / pz * fx + cx, v = p[1] / pz * fy + cy;
    r[0] = ox - u;
    r[1] = oy - v;
    if (!jac) return;
    const double ipz = 1.0 / pz;
    const double du[3] = {fx * ipz, 0.0, fx * p[0] * ipz * ipz};
    const double dv[3] = {0.0, fy * ipz, fy * p[1] * ipz * ipz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ju_w = du[0] * dpdw[0 * 3 + k] + du[1] * dpdw[1 * 3 + k] + du[2] * dpdw[2 * 3 + k];
        const double jv_w = dv[0] * dpdw[0 * 3 + k] + dv[1] * dpdw[1 * 3 + k] + dv[2] * dpdw[2 * 3 + k];
        const double ju_q = du[0] * Rm[0 * 3 + k] + du[1] * Rm[1 * 3 + k] + du[2] * Rm[2 * 3 + k];
        const double jv_q = dv[0] * Rm[0 * 3 + k] + dv[1] * Rm[1 * 3 + k] + dv[2] * Rm[2 * 3 + k];
        Jc[k] = -ju_w; Jc[6 + k] = -jv_w; Jc[3 + k] = -ju_q; Jc[9 + k] = -jv_q;
        Jp[k] = -ju_q; Jp[3 + k] = -jv_q;
    }
}

// r[2]; Jc[12] = rows (du/d[aa,t']), Jp[6]
__device__ inline void projection_residual(const double* cam, const double* X, double ox, double oy, const double* K,
                                           double r[2], double* Jc, double* Jp, bool jac) {
    const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    const double q[3] = {X[0] + cam[3], X[1] + cam[4], X[2] + cam[5]};
    double p[3], dpdw[9], Rm[9];
    angle_axis_rotate(cam, q, p, dpdw, Rm, jac);
    const double pz = p[2] * -1.0;
    const double u = p[0] / pz * fx + cx, v = p[1] / pz * fy + cy;
    r[0] = ox - u;
    r[1] = oy - v;
    if (!jac) return;
    const double ipz = 1.0 / pz;
    const double du[3] = {fx * ipz, 0.0, fx * p[0] * ipz * ipz};
    const double dv[3] = {0.0, fy * ipz, fy * p[1] * ipz * ipz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ju_w = du[0] * dpdw[0 * 3 + k] + du[1] * dpdw[1 * 3 + k] + du[2] * dpdw[2 * 3 + k];
        const double jv_w = dv[0] * dpdw[0 * 3 + k] + dv[1] * dpdw[1 * 3 + k] + dv[2] * dpdw[2 * 3 + k];
        const double ju_q = du[0] * Rm[0 * 3 + k] + du[1] * Rm[1 * 3 + k] + du[2] * Rm[2 * 3 + k];
        const double jv_q = dv[0] * Rm[0 * 3 + k] + dv[1] * Rm[1 * 3 + k] + dv[2] * Rm[2 * 3 + k];
        Jc[k] = -ju_w; Jc[6 + k] = -jv_w; Jc[3 + k] = -ju_q; Jc[9 + k] = -jv_q;
        Jp[k] = -ju_q; Jp[3 + k] = -jv_q;
    }
}

__global__ __launch_bounds__(256) void k_ba_residuals(const double* __restrict__ cams, const double* __restrict__ pts,
                                                      const double* __restrict__ obs, const int* __restrict__ cam_idx,
                                                      const int* __restrict__ pt_idx, int nobs, const double* __restrict__ K,
                                                      double* __restrict__ out_r, double* __restrict__ out_J) { BACKEND_PRIO();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nobs) return;
    double r[2], Jc[12], Jp[6];
    projection_residual(cams + 6 * cam_idx[i], pts + 3 * pt_idx[i], obs[2 * i], obs[2 * i + 1], K, r, Jc, Jp, true);
    out_r[2 * i] = r[0]; out_r[2 * i + 1] = r[1];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
#pragma unroll
        for (int k = 0; k < 6; k++) out_J[i * 18 + rr * 9 + k] = Jc[rr * 6 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) out_J[i * 18 + rr * 9 + 6 + k] = Jp[rr * 3 + k];
    }
}

// ---- deterministic block reductions -----------------------------------------------------------------------------------
// 1/sqrt(d) for d > 0: hardware estimate (v_rsq_f64) + one Newton step — ~130 clk instead of sqrt (~200) followed by a
// division (~130); accurate to a few ulp, deterministic. Used where the LM solve only needs SOME consistent Cholesky factor.
__device__ inline double rsqrt_nr(double d) {
    const double r0 = __builtin_amdgcn_rsq(d);
    const double e = 1.0 - (d * r0) * r0;
    return r0 + (0.5 * r0) * e;
}

__device__ inline double wave_max_f64(double v) {
    v = fmax(v, dpp_f64(v, 0)); v = fmax(v, dpp_f64(v, 1)); v = fmax(v, dpp_f64(v, 2)); v = fmax(v, dpp_f64(v, 3));
    return fmax(fmax(readlane_f64_c(v, 0), readlane_f64_c(v, 16)), fmax(readlane_f64_c(v, 32), readlane_f64_c(v, 48)));
}
// every thread gets the sum; red: BA_NW doubles of LDS
__device__ inline double block_sum(double v, double* red) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int i = 1; i < BA_NW; i++) s += red[i];
    return s;
}
__device__ inline double block_max(double v, double* red) {
    v = wave_max_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int i = 1; i < BA_NW; i++) s = fmax(s, red[i]);
    return s;
}

struct BAState {   // in LDS
    double x_cost, cand_cost, x_norm, radius, decrease, gmax, model_change, step_norm;
    int iter, reuse_diag, invalid, need_eval, done, termination, successful, chol_fail, first;
};

__device__ inline void huber_rho(double s, double a, double& rho0, double& rho1) {
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho0 = 2 * a * r - b;
        rho1 = fmax(DBL_MIN, a / r);
    } else { rho0 = s; rho1 = 1; }
}

// v[lane l] for a wave-uniform l (v_readlane_b32 x2)
__device__ inline double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

#define WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Camera blocks on FP64 MFMA: for camera c, F = the 2*n_c rows of its observations with columns [Jc(6) | r | 0...]; one
// wavefront accumulates D = F^T F (16x16) over K = 2*n_c in steps of 4: D[a][b] (a,b<6) = U_c, D[a][6] = rhs_c.
// The same value feeds the A and B operand of a lane (A[i=l&15][k=l>>4] = B[k=l>>4][j=l&15] = F[k][l&15]).
__device__ inline v4d cam_block_mfma(const double* __restrict__ J, const double* __restrict__ res, const int* __restrict__ cobs_list,
                                     int e0, int e1, int lane) {
    v4d acc = {0, 0, 0, 0};
    const int a = lane & 15, g = lane >> 4;
    const int nrows = __builtin_amdgcn_readfirstlane(2 * (e1 - e0));
    const bool act = a < 7;
    for (int k0 = 0; k0 < nrows; k0 += 16) {   // 4 MFMA steps per batch: all index loads, then all value loads, then the MFMAs
        int idx[4];
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = k0 + 4 * u + g;
            idx[u] = (act && r < nrows) ? cobs_list[e0 + (r >> 1)] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = k0 + 4 * u + g;
            v[u] = 0.0;
            if (idx[u] >= 0) v[u] = (a < 6) ? J[(size_t)idx[u] * 18 + (r & 1) * 6 + a] : res[2 * idx[u] + (r & 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], v[u], acc, 0, 0, 0);
    }
    return acc;   // D[row = (lane>>4) + 4*reg][col = lane&15]
}

// The whole LM solve of ONE problem in one 512-thread workgroup (round 1's kernel; see the multi-kernel chain below for why one sequence uses
// the chain). Batched form k_ba_lm_batch: blockIdx.x = problem - ONE launch per round of B sequences instead of 23, which is what the batched
// leg wants (in the mix a launch costs 60-90 us of waiting for wave slots, pmv_set_ba_mode).
__device__ __forceinline__ void ba_lm_body(const BAArgs& A) {
    __shared__ double red[BA_NW];
    __shared__ BAState st;
    __shared__ CamRot crot[32];
    __shared__ double sdj;
    extern __shared__ __attribute__((aligned(16))) double dyn[];   // M ((m+1) x m: S rows then the rhs row) | stepc (m)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nc = A.nc, np = A.np, nobs = A.nobs;
    const int n = 6 * nc + 3 * np, m = 6 * nc;
    const int ldw = A.ldw;          // number of columns of Yt / Wt (multiple of 16, >= m + 1)
    const int krows = A.krows;      // 3*np padded to a multiple of 16
    double* x = A.x; double* cand = A.cand; double* scale = A.scale; double* diag = A.diag; double* D2 = A.D2;
    double* step = A.step; double* res = A.res; double* J = A.J; double* Einv = A.Einv; double* gp = A.gp;
    double* Yt = A.Yd; double* Wt = A.Wd; double* Gp = A.Gpart;   // Yt/Wt are stored TRANSPOSED: [column][k]
    double* S = dyn; double* rhs = dyn + (size_t)m * m; double* stepc = rhs + m;

    // diagnostic phase timers (shader clock), only when A.stamps != nullptr
    unsigned long long t_prev = 0;
#define STAMP(k) do { if (A.stamps) { __syncthreads(); if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); A.stamps[k] += t_ - t_prev; t_prev = t_; } } } while (0)
    if (A.stamps && tid == 0) { t_prev = __builtin_readcyclecounter(); A.stamps[30] = wall_clock64(); A.stamps[31] = t_prev; }
    // x <- [cams | pts]; zero the K padding rows of Yt / Wt once (they are never written again)
    for (int i = tid; i < 6 * nc; i += BA_T) x[i] = A.cams[i];
    for (int i = tid; i < 3 * np; i += BA_T) x[6 * nc + i] = A.pts[i];
    {
        const int padk = krows - 3 * np;
        for (int i = tid; i < ldw * padk; i += BA_T) {
            const int col = i / padk, k = 3 * np + (i - col * padk);
            Yt[(size_t)col * krows + k] = 0.0; Wt[(size_t)col * krows + k] = 0.0;
        }
        for (int i = tid; i < (ldw - m) * 3 * np; i += BA_T) {   // Yt column m (pairs with g) and the unused tail columns
            const int col = m + i / (3 * np), k = i % (3 * np);
            Yt[(size_t)col * krows + k] = 0.0;
            if (col > m) Wt[(size_t)col * krows + k] = 0.0;
        }
    }
    if (tid == 0) {
        st.radius = 1e4; st.decrease = 2.0; st.iter = 0; st.reuse_diag = 0; st.invalid = 0; st.need_eval = 1; st.done = 0;
        st.termination = 0; st.successful = 0; st.chol_fail = 0; st.first = 1; st.gmax = 0; st.x_cost = 0;
    }
    __syncthreads();

    // camera blocks (all cameras, waves in parallel) into S / rhs; S must have been zeroed
    auto camera_blocks = [&]() {
        for (int c = wid; c < nc; c += BA_NW) {
            const v4d d = cam_block_mfma(J, res, A.cobs_list, A.cobs_start[c], A.cobs_start[c + 1], lane);
            const int col = lane & 15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = (lane >> 4) + 4 * r;
                if (row < 6) {
                    if (col < 6) S[(size_t)(6 * c + row) * m + 6 * c + col] = d[r];
                    else if (col == 6) rhs[6 * c + row] = d[r];
                }
            }
        }
    };

    for (;;) {
        // ================= (re-)evaluate cost, corrected residuals and Jacobians at x ======================================
        if (st.need_eval) {
            double cpart = 0;
            if (tid < nc) cam_rot_setup(x + 6 * tid, crot[tid]);
            __syncthreads();
            for (int i = tid; i < nobs; i += BA_T) {
                double r[2], Jc[12], Jp[6];
                const int c = A.cam_idx[i], p = A.pt_idx[i];
                projection_residual_pre(crot[c], x + 6 * c, x + 6 * nc + 3 * p, A.obs[2 * i], A.obs[2 * i + 1], A.K, r, Jc, Jp, true);
                double rho0, rho1;
                huber_rho(r[0] * r[0] + r[1] * r[1], A.huber, rho0, rho1);
                cpart += 0.5 * rho0;
                const double sr = sqrt(rho1);
                res[2 * i] = r[0] * sr; res[2 * i + 1] = r[1] * sr;
                if (st.first) {
#pragma unroll
                    for (int k = 0; k < 12; k++) J[(size_t)i * 18 + k] = Jc[k] * sr;
#pragma unroll
                    for (int k = 0; k < 6; k++) J[(size_t)i * 18 + 12 + k] = Jp[k] * sr;
                } else {   // Jacobi scaling is fixed after the first evaluation
                    const double* sc = scale + 6 * c;
                    const double* sp = scale + 6 * nc + 3 * p;
#pragma unroll
                    for (int k = 0; k < 12; k++) J[(size_t)i * 18 + k] = Jc[k] * sr * sc[k % 6];
#pragma unroll
                    for (int k = 0; k < 6; k++) J[(size_t)i * 18 + 12 + k] = Jp[k] * sr * sp[k % 3];
                }
            }
            const double xc = block_sum(cpart, red);
            double xn = 0;
            for (int i = tid; i < n; i += BA_T) xn += x[i] * x[i];
            xn = block_sum(xn, red);
            if (tid == 0) { st.x_cost = xc; st.x_norm = sqrt(xn); if (st.first) A.summary[0] = xc; }
            __syncthreads();
            if (st.first) {
                // jacobian column norms -> scale = 1/(1+||col||): cameras from the diagonal of the (unscaled) camera blocks,
                // points from their observation lists
                camera_blocks();
                __syncthreads();
                for (int i = tid; i < m; i += BA_T) scale[i] = 1.0 / (1.0 + sqrt(S[(size_t)i * m + i]));
                for (int p = tid; p < np; p += BA_T) {
                    double acc[3] = {0, 0, 0};
                    for (int e = A.pobs_start[p]; e < A.pobs_start[p + 1]; e++) {
                        const double* Jp = J + (size_t)A.pobs_list[e] * 18 + 12;
#pragma unroll
                        for (int k = 0; k < 3; k++) acc[k] += Jp[k] * Jp[k] + Jp[3 + k] * Jp[3 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) scale[6 * nc + 3 * p + k] = 1.0 / (1.0 + sqrt(acc[k]));
                }
                __syncthreads();
                for (int i = tid; i < nobs; i += BA_T) {
                    const double* sc = scale + 6 * A.cam_idx[i];
                    const double* sp = scale + 6 * nc + 3 * A.pt_idx[i];
#pragma unroll
                    for (int k = 0; k < 12; k++) J[(size_t)i * 18 + k] *= sc[k % 6];
#pragma unroll
                    for (int k = 0; k < 6; k++) J[(size_t)i * 18 + 12 + k] *= sp[k % 3];
                }
            }
            if (tid == 0) { st.need_eval = 0; st.first = 0; st.gmax = -1.0; }
            __syncthreads();
        }
        STAMP(0);
        // ================= loop-top termination tests (FinalizeIterationAndCheckIfMinimizerCanContinue) ======================
        // (gmax is produced by the Schur phase below; on the first pass it is not known yet and is checked after that phase)
        if (tid == 0) {
            if (st.iter >= A.max_iterations) { st.done = 1; st.termination = 0; }
            else if (st.gmax >= 0 && st.gmax <= 1e-10) { st.done = 1; st.termination = 2; }
            else if (st.radius < 1e-32) { st.done = 1; st.termination = 4; }
        }
        __syncthreads();
        if (st.done) break;

        // ================= camera blocks: U_c = sum Jc^T Jc, rhs_c = sum Jc^T r (one MFMA wavefront per camera) ==============
        for (int i = tid; i < m * m; i += BA_T) S[i] = 0.0;
        __syncthreads();
        camera_blocks();
        __syncthreads();
        STAMP(1);
        // ================= LM diagonal ======================================================================================
        if (!st.reuse_diag) {
            for (int i = tid; i < m; i += BA_T) diag[i] = fmin(fmax(S[(size_t)i * m + i], 1e-6), 1e32);
            for (int p = tid; p < np; p += BA_T) {
                double acc[3] = {0, 0, 0};
                for (int e = A.pobs_start[p]; e < A.pobs_start[p + 1]; e++) {
                    const double* Jp = J + (size_t)A.pobs_list[e] * 18 + 12;
#pragma unroll
                    for (int k = 0; k < 3; k++) acc[k] += Jp[k] * Jp[k] + Jp[3 + k] * Jp[3 + k];
                }
#pragma unroll
                for (int k = 0; k < 3; k++) diag[m + 3 * p + k] = fmin(fmax(acc[k], 1e-6), 1e32);
            }
        }
        __syncthreads();
        {
            const double radius = st.radius;
            for (int i = tid; i < n; i += BA_T) D2[i] = diag[i] / radius;
        }
        __syncthreads();
        for (int i = tid; i < m; i += BA_T) S[(size_t)i * m + i] += D2[i];
        STAMP(2);
        // ================= point blocks: E^-1, g, and the point's three K-columns of Yt and [Wt | g] =========================
        if (tid == 0) st.chol_fail = 0;
        __syncthreads();
        double gmax_p = 0;
        for (int p = tid; p < np; p += BA_T) {
            double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gv[3] = {0, 0, 0};
            unsigned seen = 0;
            for (int e = A.pobs_start[p]; e < A.pobs_start[p + 1]; e++) {
                const int i = A.pobs_list[e];
                seen |= 1u << A.cam_idx[i];
                const double* Jp = J + (size_t)i * 18 + 12;
                const double r0 = res[2 * i], r1 = res[2 * i + 1];
#pragma unroll
                for (int a = 0; a < 3; a++) {
#pragma unroll
                    for (int b = 0; b < 3; b++) E[a * 3 + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
                    gv[a] += Jp[a] * r0 + Jp[3 + a] * r1;
                }
            }
#pragma unroll
            for (int a = 0; a < 3; a++) {
                E[a * 3 + a] += D2[m + 3 * p + a];
                gmax_p = fmax(gmax_p, fabs(gv[a] / scale[m + 3 * p + a]));
            }
            // 3x3 Cholesky inverse
            double L[9];
            bool ok = true;
            {
                double d = E[0];
                ok = ok && (d > 0.0); L[0] = sqrt(d);
                L[3] = E[3] / L[0]; L[6] = E[6] / L[0];
                d = E[4] - L[3] * L[3];
                ok = ok && (d > 0.0); L[4] = sqrt(d);
                L[7] = (E[7] - L[6] * L[3]) / L[4];
                d = E[8] - L[6] * L[6] - L[7] * L[7];
                ok = ok && (d > 0.0); L[8] = sqrt(d);
            }
            if (!ok) { st.chol_fail = 1; continue; }
            double Ei[9];
#pragma unroll
            for (int cI = 0; cI < 3; cI++) {
                double e0 = (cI == 0) ? 1.0 : 0.0, e1 = (cI == 1) ? 1.0 : 0.0, e2 = (cI == 2) ? 1.0 : 0.0;
                e0 = e0 / L[0];
                e1 = (e1 - L[3] * e0) / L[4];
                e2 = (e2 - L[6] * e0 - L[7] * e1) / L[8];
                e2 = e2 / L[8];
                e1 = (e1 - L[7] * e2) / L[4];
                e0 = (e0 - L[6] * e2 - L[3] * e1) / L[0];
                Ei[0 * 3 + cI] = e0; Ei[1 * 3 + cI] = e1; Ei[2 * 3 + cI] = e2;
            }
#pragma unroll
            for (int k = 0; k < 9; k++) Einv[(size_t)p * 9 + k] = Ei[k];
#pragma unroll
            for (int k = 0; k < 3; k++) { gp[(size_t)p * 3 + k] = gv[k]; Wt[(size_t)m * krows + 3 * p + k] = gv[k]; }
            // cameras that do not see this point: zero K-columns (adjacent points -> adjacent addresses, coalesced)
            for (int c = 0; c < nc; c++) {
                if (seen & (1u << c)) continue;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        Wt[(size_t)(6 * c + a) * krows + 3 * p + b] = 0.0;
                        Yt[(size_t)(6 * c + a) * krows + 3 * p + b] = 0.0;
                    }
                }
            }
            for (int e = A.pobs_start[p]; e < A.pobs_start[p + 1]; e++) {
                const int i = A.pobs_list[e];
                const int c = A.cam_idx[i];
                bool dup = false;
                for (int e2 = A.pobs_start[p]; e2 < e; e2++) dup = dup || (A.cam_idx[A.pobs_list[e2]] == c);
                const double* Jc = J + (size_t)i * 18;
                const double* Jp = Jc + 12;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    double w3[3];
#pragma unroll
                    for (int b = 0; b < 3; b++) w3[b] = Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b];
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        const double y = w3[0] * Ei[0 * 3 + b] + w3[1] * Ei[1 * 3 + b] + w3[2] * Ei[2 * 3 + b];
                        double* wp = &Wt[(size_t)(6 * c + a) * krows + 3 * p + b];
                        double* yp = &Yt[(size_t)(6 * c + a) * krows + 3 * p + b];
                        if (dup) { *wp += w3[b]; *yp += y; }      // two features of one frame on the same landmark (rare)
                        else { *wp = w3[b]; *yp = y; }
                    }
                }
            }
        }
        STAMP(3);
        // gradient max norm of the unscaled problem (cameras from rhs, points from g)
        for (int i = tid; i < m; i += BA_T) gmax_p = fmax(gmax_p, fabs(rhs[i] / scale[i]));
        const double gm = block_max(gmax_p, red);
        if (tid == 0 && st.gmax < 0) st.gmax = gm;
        __syncthreads();
        if (st.gmax <= 1e-10) { if (tid == 0) { st.done = 1; st.termination = 2; } __syncthreads(); break; }

        bool valid = !st.chol_fail;
        if (valid) {
            STAMP(4);
            // ================= Schur complement on FP64 MFMA: G = Y^T [W | g]  (m_pad x ncol_pad, K = krows) ================
            // Operands are K-contiguous ([column][k]); a lane fetches 4 consecutive k (32 B) per operand and 16 k are consumed
            // by 4 MFMAs: in MFMA s, lane group g supplies k = k0 + 4g + s for both operands (any pairing of k is a valid
            // order of the sum).
            const int tr = A.tiles_r, tc = A.tiles_c, ks = A.kslices, kper = A.kper;
            const int items = tr * tc * ks;
            for (int it = wid; it < items; it += BA_NW) {
                const int s = it / (tr * tc), tile = it - s * (tr * tc);
                const int ti = tile / tc, tj = tile - ti * tc;
                const int k0 = s * kper, k1 = min(krows, k0 + kper);
                v4d acc = {0, 0, 0, 0};
                const double* ya = Yt + (size_t)(ti * 16 + (lane & 15)) * krows + 4 * (lane >> 4);
                const double* wb = Wt + (size_t)(tj * 16 + (lane & 15)) * krows + 4 * (lane >> 4);
                const int nk = __builtin_amdgcn_readfirstlane((k1 - k0) / 16);
                const double* pa = ya + k0;
                const double* pb = wb + k0;
                int kb = 0;
                for (; kb + 4 <= nk; kb += 4) {   // 4 K-blocks (64 k): 8 x 32-byte loads per lane in flight, then 16 MFMAs
                    v4d a4[4], b4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { a4[u] = *(const v4d*)(pa + 16 * u); b4[u] = *(const v4d*)(pb + 16 * u); }
                    pa += 64; pb += 64;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
#pragma unroll
                        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[u][q], b4[u][q], acc, 0, 0, 0);
                    }
                }
                for (; kb < nk; kb++) {
                    const v4d a4 = *(const v4d*)pa, b4 = *(const v4d*)pb;
                    pa += 16; pb += 16;
#pragma unroll
                    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[q], b4[q], acc, 0, 0, 0);
                }
                // D[row = (lane>>4) + 4*reg][col = lane&15]
#pragma unroll
                for (int r = 0; r < 4; r++)
                    Gp[((size_t)s * A.gp_rows + ti * 16 + (lane >> 4) + 4 * r) * ldw + tj * 16 + (lane & 15)] = acc[r];
            }
            __syncthreads();
            for (int i = tid; i < m * (m + 1); i += BA_T) {
                const int a = i / (m + 1), b = i - a * (m + 1);
                double g = 0;
                for (int s = 0; s < ks; s++) g += Gp[((size_t)s * A.gp_rows + a) * ldw + b];
                if (b < m) S[(size_t)a * m + b] -= g;
                else rhs[a] -= g;
            }
            __syncthreads();
            STAMP(5);
            // ================= Cholesky of S with the rhs carried as row m (forward substitution for free) ==================
            // right-looking; element (i,k) receives its updates in ascending j — the order of a left-looking factorisation
            // and of a row-oriented forward substitution. All 8 wavefronts share the trailing update.
            for (int j = 0; j < m; j++) {
                if (tid == 0) {
                    const double d = S[(size_t)j * m + j];
                    if (!(d > 0.0)) st.chol_fail = 1;
                    else { sdj = sqrt(d); S[(size_t)j * m + j] = sdj; }
                }
                __syncthreads();
                if (st.chol_fail) break;
                const double dj = sdj;
                for (int i = j + 1 + tid; i <= m; i += BA_T) S[(size_t)i * m + j] /= dj;   // row m = rhs row
                __syncthreads();
                const int rem = m - j - 1;   // columns j+1 .. m-1; rows j+1 .. m
                for (int e = tid; e < (rem + 1) * rem; e += BA_T) {
                    const int i = j + 1 + e / rem, k = j + 1 + e % rem;
                    if (k <= i) S[(size_t)i * m + k] -= S[(size_t)i * m + j] * S[(size_t)k * m + j];
                }
                __syncthreads();
            }
            valid = !st.chol_fail;
            if (valid && wid == 0) {
                // backward substitution on y = row m (column-oriented, descending), wave-synchronous
                for (int i = lane; i < m; i += 64) stepc[i] = rhs[i];
                WAVE_SYNC();
                for (int i = m - 1; i >= 0; i--) {
                    const double xi = stepc[i] / S[(size_t)i * m + i];
                    WAVE_SYNC();
                    for (int k = lane; k < m; k += 64) {
                        if (k == i) stepc[i] = xi;
                        else if (k < i) stepc[k] -= S[(size_t)i * m + k] * xi;
                    }
                    WAVE_SYNC();
                }
            }
            __syncthreads();
        }
        if (valid) {
            STAMP(6);
            for (int i = tid; i < m; i += BA_T) step[i] = stepc[i];
            __syncthreads();
            // point back-substitution: y_p = E^-1 (g_p - sum W^T y_c)
            for (int p = tid; p < np; p += BA_T) {
                double t3[3] = {gp[(size_t)p * 3], gp[(size_t)p * 3 + 1], gp[(size_t)p * 3 + 2]};
                for (int e = A.pobs_start[p]; e < A.pobs_start[p + 1]; e++) {
                    const int i = A.pobs_list[e];
                    const int c = A.cam_idx[i];
                    const double* Jc = J + (size_t)i * 18;
                    const double* Jp = Jc + 12;
                    double jy0 = 0, jy1 = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) { jy0 += Jc[a] * step[6 * c + a]; jy1 += Jc[6 + a] * step[6 * c + a]; }
#pragma unroll
                    for (int a = 0; a < 3; a++) t3[a] -= Jp[a] * jy0 + Jp[3 + a] * jy1;
                }
                const double* Ei = Einv + (size_t)p * 9;
#pragma unroll
                for (int a = 0; a < 3; a++) step[m + 3 * p + a] = Ei[a * 3] * t3[0] + Ei[a * 3 + 1] * t3[1] + Ei[a * 3 + 2] * t3[2];
            }
            __syncthreads();
            for (int i = tid; i < n; i += BA_T) step[i] = -step[i];
            __syncthreads();
            STAMP(7);
            // model cost change = -(J step)^T (r + J step / 2)
            double mc = 0;
            for (int i = tid; i < nobs; i += BA_T) {
                const int c = A.cam_idx[i], p = A.pt_idx[i];
                const double* Jr = J + (size_t)i * 18;
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    double mr = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) mr += Jr[rr * 6 + k] * step[6 * c + k];
#pragma unroll
                    for (int k = 0; k < 3; k++) mr += Jr[12 + rr * 3 + k] * step[m + 3 * p + k];
                    mc -= mr * (res[2 * i + rr] + mr / 2.0);
                }
            }
            mc = block_sum(mc, red);
            if (tid == 0) st.model_change = mc;
            __syncthreads();
            valid = st.model_change > 0.0;
        }
        if (tid == 0) st.iter++;
        if (!valid) {   // HandleInvalidStep
            if (tid == 0) {
                if (++st.invalid >= 5) { st.done = 1; st.termination = 4; }
                else { st.radius /= st.decrease; st.decrease *= 2; st.reuse_diag = 1; }
            }
            __syncthreads();
            if (st.done) break;
            continue;
        }
        STAMP(8);
        // ================= candidate, tolerances, accept / reject =============================================================
        double sn = 0;
        for (int i = tid; i < n; i += BA_T) {
            const double d = step[i] * scale[i];
            cand[i] = x[i] + d;
            sn += d * d;
        }
        sn = block_sum(sn, red);
        double cpart = 0;
        if (tid < nc) cam_rot_setup(cand + 6 * tid, crot[tid]);
        __syncthreads();
        for (int i = tid; i < nobs; i += BA_T) {
            double r[2];
            const int c = A.cam_idx[i], p = A.pt_idx[i];
            projection_residual_pre(crot[c], cand + 6 * c, cand + 6 * nc + 3 * p, A.obs[2 * i], A.obs[2 * i + 1], A.K, r, nullptr, nullptr, false);
            double rho0, rho1;
            huber_rho(r[0] * r[0] + r[1] * r[1], A.huber, rho0, rho1);
            cpart += 0.5 * rho0;
        }
        const double cc = block_sum(cpart, red);
        if (tid == 0) {
            st.invalid = 0;
            st.cand_cost = cc;
            st.step_norm = sqrt(sn);
            const double cost_change = st.x_cost - cc;
            if (st.step_norm <= 1e-8 * (st.x_norm + 1e-8)) { st.done = 1; st.termination = 3; }
            else if (fabs(cost_change) <= 1e-6 * st.x_cost) { st.done = 1; st.termination = 1; }
            else {
                const double rel = cost_change / st.model_change;
                if (rel > 1e-3) {
                    st.need_eval = 2;   // accept: x <- cand below
                    st.successful++;
                    const double t = 2.0 * rel - 1.0;
                    st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                    st.radius = fmin(1e16, st.radius);
                    st.decrease = 2.0;
                    st.reuse_diag = 0;
                } else {
                    st.radius /= st.decrease; st.decrease *= 2; st.reuse_diag = 1;
                }
            }
        }
        __syncthreads();
        if (st.done) break;
        STAMP(9);
        if (st.need_eval == 2) {
            for (int i = tid; i < n; i += BA_T) x[i] = cand[i];
            __syncthreads();
            if (tid == 0) {
                if (st.iter >= A.max_iterations) {
                    // last permitted iteration accepted: the re-evaluation at the new x would only reproduce cand_cost
                    st.x_cost = st.cand_cost; st.need_eval = 0; st.done = 1; st.termination = 0;
                } else st.need_eval = 1;
            }
            __syncthreads();
            if (st.done) break;
        }
    }
    __syncthreads();
    for (int i = tid; i < 6 * nc; i += BA_T) A.cams[i] = x[i];
    for (int i = tid; i < 3 * np; i += BA_T) A.pts[i] = x[6 * nc + i];
    if (A.stamps && tid == 0) { A.stamps[28] = wall_clock64() - A.stamps[30]; A.stamps[29] = __builtin_readcyclecounter() - A.stamps[31]; }
    if (tid == 0) {
        A.summary[1] = st.x_cost; A.summary[2] = st.iter; A.summary[3] = st.successful; A.summary[4] = st.termination;
    }
}
__global__ __launch_bounds__(BA_T) void k_ba_lm(BAArgs A) { BACKEND_PRIO(); ba_lm_body(A); }
// args[blockIdx.x] = one problem; when A.out is set (mapped pinned result block: [summary 8 | cams | pts]) the result is copied there
__global__ __launch_bounds__(BA_T) void k_ba_lm_batch(const BAArgs* __restrict__ args) { BACKEND_PRIO();
    const BAArgs A = args[blockIdx.x];
    ba_lm_body(A);
    if (A.out) {
        __syncthreads();   // (the body's last stores to A.cams / A.pts / A.summary: same workgroup)
        const int nsum = 8, ncam = 6 * A.nc, npt = 3 * A.np;
        for (int i = threadIdx.x; i < nsum + ncam + npt; i += BA_T)
            A.out[i] = i < nsum ? A.summary[i] : (i < nsum + ncam ? A.cams[i - nsum] : A.pts[i - nsum - ncam]);
    }
}

// =========================================================================================================================
// Multi-kernel LM (the default path): the same iteration as k_ba_lm with every data-parallel phase spread over many CUs —
// one CU moves only ~10 B/clk and an FP64 dependency chain costs ~18 clk/op, so a single workgroup is bound by its own
// bandwidth and latency. All decisions stay on the device: the host enqueues a fixed chain of launches
//     E (accept/reject of the previous step + r, J at the new point)  ->  C|P (camera blocks on MFMA | point blocks)  ->
//     G (Schur contraction on MFMA, one wavefront per tile x K-slice)  ->  S (reduced system, Cholesky)  ->  B (back-
//     substitution, candidate, model-cost and candidate-cost terms)
// per LM iteration; every kernel reads the device-side state and returns at once when the solve has terminated. Kernel
// boundaries are the only grid-wide synchronisation (no cooperative launch, no spinning), all sums have a fixed order.
// The state is double-buffered so that E can evaluate the decision redundantly in every block while block 0 publishes it.
// =========================================================================================================================
struct BAGState {
    double x_cost, cand_cost, x_norm, radius, decrease, gmax, model_change, step_norm, initial_cost;
    int iter, reuse_diag, invalid, need_eval, done, termination, successful, chol_fail, first, cur, step_valid, max_iterations;
};
constexpr int BM_T = 256;          // threads per block of the E / C|P / S / B kernels
constexpr int BM_NW = BM_T / 64;

// accept / reject, trust-region update, termination (ceres TrustRegionMinimizer) on a private copy of the state
__device__ inline void bam_decide(BAGState& s, const double* __restrict__ part4, int nbp) {
    s.iter++;
    s.first = 0;
    bool valid = s.step_valid != 0;
    double mc = 0, cc = 0, dn2 = 0;
    if (valid) {
        double xn2 = 0;
        for (int b = 0; b < nbp; b++) { mc += part4[b * 4]; cc += part4[b * 4 + 1]; dn2 += part4[b * 4 + 2]; xn2 += part4[b * 4 + 3]; }
        s.x_norm = sqrt(xn2);   // |x| of the point the step was taken from (summed by the back-substitution kernel)
        s.model_change = mc;
        valid = mc > 0.0;
    }
    s.chol_fail = 0;
    if (!valid) {   // HandleInvalidStep
        if (++s.invalid >= 5) { s.done = 1; s.termination = 4; }
        else { s.radius /= s.decrease; s.decrease *= 2; s.reuse_diag = 1; }
        s.need_eval = 0;
        return;
    }
    s.invalid = 0;
    s.cand_cost = cc;
    s.step_norm = sqrt(dn2);
    const double cost_change = s.x_cost - cc;
    if (s.step_norm <= 1e-8 * (s.x_norm + 1e-8)) { s.done = 1; s.termination = 3; return; }
    if (fabs(cost_change) <= 1e-6 * s.x_cost) { s.done = 1; s.termination = 1; return; }
    const double rel = cost_change / mc;
    if (rel > 1e-3) {
        s.cur ^= 1;
        s.successful++;
        const double t = 2.0 * rel - 1.0;
        s.radius = s.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        s.radius = fmin(1e16, s.radius);
        s.decrease = 2.0;
        s.reuse_diag = 0;
        s.need_eval = 1;
        s.x_cost = cc;   // the accepted candidate's cost (its r and J were evaluated by the back-substitution kernel)
        if (s.iter >= s.max_iterations) { s.done = 1; s.termination = 0; }
    } else {
        s.radius /= s.decrease; s.decrease *= 2; s.reuse_diag = 1;
        s.need_eval = 0;
    }
}

// E0 (first iteration only): initial state, x <- caller's cameras and points, r and J (Huber-corrected, unscaled: the Jacobi
// scaling is derived from these columns) into buffer 0; one thread per observation; cost partial per block. From the second
// iteration on the back-substitution kernel evaluates r and J at the candidate, so an accepted step needs no extra pass.
__device__ inline void bam_eval0_role(const BAArgs& A, BAGState* __restrict__ st_out, int* __restrict__ chol_flags,
                                      double* __restrict__ part_cost, int bid, int nblk) {
    __shared__ double red[BM_NW];
    const int tid = threadIdx.x;
    const int n = 6 * A.nc + 3 * A.np;
    if (bid == 0) {
        if (tid == 0) {
            BAGState ss;
            ss.radius = 1e4; ss.decrease = 2.0; ss.iter = 0; ss.reuse_diag = 0; ss.invalid = 0; ss.need_eval = 1; ss.done = 0;
            ss.termination = 0; ss.successful = 0; ss.chol_fail = 0; ss.first = 1; ss.cur = 0; ss.step_valid = 0;
            ss.gmax = -1.0; ss.x_cost = 0; ss.initial_cost = 0; ss.max_iterations = A.max_iterations; ss.x_norm = 0;
            ss.cand_cost = 0; ss.model_change = 0; ss.step_norm = 0;
            *st_out = ss;
        }
        for (int i = tid; i < A.max_iterations; i += BM_T) chol_flags[i] = 0;
    }
    for (int i = bid * BM_T + tid; i < n; i += nblk * BM_T) A.x[i] = (i < 6 * A.nc) ? A.cams[i] : A.pts[i - 6 * A.nc];
    const int i = bid * BM_T + tid;
    double cpart = 0;
    if (i < A.nobs) {
        const int c = A.cam_idx[i], p = A.pt_idx[i];
        CamRot cr;
        cam_rot_setup(A.cams + 6 * c, cr);
        double r[2], Jc[12], Jp[6];
        projection_residual_pre(cr, A.cams + 6 * c, A.pts + 3 * p, A.obs[2 * i], A.obs[2 * i + 1], A.K, r, Jc, Jp, true);
        double rho0, rho1;
        huber_rho(r[0] * r[0] + r[1] * r[1], A.huber, rho0, rho1);
        cpart = 0.5 * rho0;
        const double sr = sqrt(rho1);
        A.res[2 * i] = r[0] * sr; A.res[2 * i + 1] = r[1] * sr;
        double* Jo = A.J + (size_t)i * 18;
#pragma unroll
        for (int k = 0; k < 12; k++) Jo[k] = Jc[k] * sr;
#pragma unroll
        for (int k = 0; k < 6; k++) Jo[12 + k] = Jp[k] * sr;
    }
    cpart = wave_sum_f64(cpart);
    if ((tid & 63) == 0) red[tid >> 6] = cpart;
    __syncthreads();
    if (tid == 0) part_cost[bid] = (red[0] + red[1]) + (red[2] + red[3]);
}

// blocks [0, eval_blocks): E0; the remaining blocks clear Yt | [Wt | g] (adjacent; columns of cameras that do not see a point are
// never written by the point kernel) — the clear rides along instead of being a launch of its own
__global__ __launch_bounds__(BM_T) void k_bam_eval0(BAArgs A, BAGState* st_out, int* chol_flags, double* part_cost, int eval_blocks) { BACKEND_PRIO();
    if ((int)blockIdx.x < eval_blocks) { bam_eval0_role(A, st_out, chol_flags, part_cost, blockIdx.x, eval_blocks); return; }
    const size_t n2 = (size_t)A.krows * A.ldw;   // doubles in Yt + Wt = 2 * n2, a multiple of 2
    double2* z = (double2*)A.Yd;
    const int zb = blockIdx.x - eval_blocks, nzb = gridDim.x - eval_blocks;
    for (size_t i = (size_t)zb * BM_T + threadIdx.x; i < n2; i += (size_t)nzb * BM_T) z[i] = double2{0.0, 0.0};
}

// C role: one block (4 wavefronts) per camera: U_c (6x6) and rhs_c on FP64 MFMA; in the first iteration also the Jacobi
// scale of the camera's parameters (from the unscaled column norms = diag U_c) and the rescaled block.
__device__ inline void bam_cam_role(const BAArgs& A, const BAGState& st, int c, double* __restrict__ Ublk,
                                    double* __restrict__ rhsblk, double* sred /* [BM_NW][64][4] */, double* ssc /* 8 */) {
    if (st.done || !st.need_eval) return;
    const double* Jb = A.J + (size_t)st.cur * A.nobs * 18;
    const double* resb = A.res + (size_t)st.cur * A.nobs * 2;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int e0 = A.cobs_start[c], e1 = A.cobs_start[c + 1];
    const int per = (((e1 - e0) + BM_NW - 1) / BM_NW + 7) & ~7;   // observations per wavefront, whole MFMA batches (8 obs)
    const int w0 = min(e1, e0 + wid * per), w1 = min(e1, w0 + per);
    const v4d d = cam_block_mfma(Jb, resb, A.cobs_list, w0, w1, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) sred[(wid * 4 + r) * 64 + lane] = d[r];
    __syncthreads();
    if (wid != 0) return;
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = (sred[(0 * 4 + r) * 64 + lane] + sred[(1 * 4 + r) * 64 + lane]) + (sred[(2 * 4 + r) * 64 + lane] + sred[(3 * 4 + r) * 64 + lane]);
    const int col = lane & 15;
    const bool first = st.first != 0;
    if (first) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (lane >> 4) + 4 * r;
            if (row < 6 && col == row) { const double sc = 1.0 / (1.0 + sqrt(v[r])); ssc[row] = sc; A.scale[6 * c + row] = sc; }
        }
        WAVE_SYNC();
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = (lane >> 4) + 4 * r;
        if (row < 6 && col < 7) {
            double o = v[r];
            if (first) o = (col < 6) ? o * ssc[row] * ssc[col] : o * ssc[row];
            if (col < 6) Ublk[c * 36 + row * 6 + col] = o;
            else rhsblk[c * 6 + row] = o;
        }
    }
}

// P role: one block per BM_PB points. Phase 1, one thread per point: (first iteration: Jacobi scale of the point and
// scaling of the Jacobian rows of its observations), E, g, LM diagonal, E^-1 (LDS + global). Phase 2, one thread per
// observation of the block's points: the observation's 6x3 blocks of Wt and Yt = W E^-1. Gradient-max partial per block.
constexpr int BM_OB = 8;    // observations of a point handled per register batch
constexpr int BM_PB = 64;   // points per block
__device__ inline void bam_point_role(const BAArgs& A, const BAGState& st, int* __restrict__ chol_flag, int pb,
                                      double* __restrict__ part_gmax, double* sEi /* [BM_PB][9] */) {
    if (st.done) return;
    unsigned long long t_prev = __builtin_readcyclecounter();
#define PSTAMP(k) do { if (A.stamps && pb == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); A.stamps[k] += t_ - t_prev; t_prev = t_; } } while (0)
    double* Jb = A.J + (size_t)st.cur * A.nobs * 18;
    const double* resb = A.res + (size_t)st.cur * A.nobs * 2;
    const int m = 6 * A.nc, krows = A.krows;
    const int tid = threadIdx.x;
    const int p0 = pb * BM_PB, p1 = min(A.np, p0 + BM_PB);
    const int p = p0 + tid;
    double* sSp = sEi + BM_PB * 9;   // [BM_PB][3] point scales (first iteration)
    double gmax_p = 0;
    if (tid < BM_PB && p < p1) {
        const int e0 = A.pobs_start[p], e1 = A.pobs_start[p + 1];
        double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gv[3] = {0, 0, 0};
        for (int eb = e0; eb < e1; eb += BM_OB) {   // index loads, then value loads, then arithmetic: independent loads in flight
            int oi[BM_OB];
#pragma unroll
            for (int u = 0; u < BM_OB; u++) oi[u] = (eb + u < e1) ? A.pobs_list[eb + u] : -1;
            double jp[BM_OB][6], rr[BM_OB][2];
#pragma unroll
            for (int u = 0; u < BM_OB; u++) {
                if (oi[u] >= 0) {
                    const double* Jp = Jb + (size_t)oi[u] * 18 + 12;
#pragma unroll
                    for (int k = 0; k < 6; k++) jp[u][k] = Jp[k];
                    rr[u][0] = resb[2 * oi[u]]; rr[u][1] = resb[2 * oi[u] + 1];
                } else {
#pragma unroll
                    for (int k = 0; k < 6; k++) jp[u][k] = 0;
                    rr[u][0] = rr[u][1] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < BM_OB; u++) {
                if (oi[u] >= 0) {
#pragma unroll
                    for (int a = 0; a < 3; a++) {
#pragma unroll
                        for (int b = 0; b < 3; b++) E[a * 3 + b] += jp[u][a] * jp[u][b] + jp[u][3 + a] * jp[u][3 + b];
                        gv[a] += jp[u][a] * rr[u][0] + jp[u][3 + a] * rr[u][1];
                    }
                }
            }
        }
        if (st.first) {
            // Jacobi scaling of the point's columns: their squared norms are the diagonal of the unscaled E; E and g of the
            // scaled problem follow by scaling (the Jacobian rows themselves are rescaled by the per-observation pass below)
            double sp[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { sp[k] = 1.0 / (1.0 + sqrt(E[k * 3 + k])); A.scale[m + 3 * p + k] = sp[k]; sSp[tid * 3 + k] = sp[k]; }
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int b = 0; b < 3; b++) E[a * 3 + b] = E[a * 3 + b] * sp[a] * sp[b];
                gv[a] = gv[a] * sp[a];
            }
        }
        const double radius = st.radius;
        const bool reuse = st.reuse_diag != 0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            double dg;
            if (!reuse) { dg = fmin(fmax(E[a * 3 + a], 1e-6), 1e32); A.diag[m + 3 * p + a] = dg; }
            else dg = A.diag[m + 3 * p + a];
            E[a * 3 + a] += dg / radius;
            gmax_p = fmax(gmax_p, fabs(gv[a] / A.scale[m + 3 * p + a]));
        }
        double L[9];
        bool ok = true;
        double r0 = 0, r1 = 0, r2 = 0;   // reciprocals of the diagonal of L (rsqrt + Newton: no division on the chain)
        {
            double d = E[0];
            ok = ok && (d > 0.0); r0 = rsqrt_nr(ok ? d : 1.0);
            L[3] = E[3] * r0; L[6] = E[6] * r0;
            d = E[4] - L[3] * L[3];
            ok = ok && (d > 0.0); r1 = rsqrt_nr(ok ? d : 1.0);
            L[7] = (E[7] - L[6] * L[3]) * r1;
            d = E[8] - L[6] * L[6] - L[7] * L[7];
            ok = ok && (d > 0.0); r2 = rsqrt_nr(ok ? d : 1.0);
        }
        double Ei[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (!ok) atomicOr(chol_flag, 1);   // the step is invalid: the solve kernel stops, nothing below is used
        else {
#pragma unroll
            for (int cI = 0; cI < 3; cI++) {
                double q0 = (cI == 0) ? 1.0 : 0.0, q1 = (cI == 1) ? 1.0 : 0.0, q2 = (cI == 2) ? 1.0 : 0.0;
                q0 = q0 * r0;
                q1 = (q1 - L[3] * q0) * r1;
                q2 = (q2 - L[6] * q0 - L[7] * q1) * r2;
                q2 = q2 * r2;
                q1 = (q1 - L[7] * q2) * r1;
                q0 = (q0 - L[6] * q2 - L[3] * q1) * r0;
                Ei[0 * 3 + cI] = q0; Ei[1 * 3 + cI] = q1; Ei[2 * 3 + cI] = q2;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; k++) { A.Einv[(size_t)p * 9 + k] = Ei[k]; sEi[tid * 9 + k] = Ei[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { A.gp[(size_t)p * 3 + k] = gv[k]; A.Wd[(size_t)m * krows + 3 * p + k] = gv[k]; }
    }
    gmax_p = wave_max_f64(gmax_p);
    if (tid == 0) part_gmax[pb] = gmax_p;   // the points of the block live in wavefront 0
    PSTAMP(4);
    __syncthreads();                        // E^-1 (and, first iteration, the point scales) in LDS
    PSTAMP(5);
    const int eb0 = A.pobs_start[p0], eb1 = A.pobs_start[p1];
    if (st.first) {   // block-uniform: one thread per observation rescales its Jacobian row in place (cameras: scales of the C kernel)
        for (int e = eb0 + tid; e < eb1; e += BM_T) {
            const int4 rec = A.erec[e];
            const int i = rec.x;
            const double* sc = A.scale + 6 * rec.y;
            const double* sp = sSp + (rec.z - p0) * 3;
            double* Jo = Jb + (size_t)i * 18;
#pragma unroll
            for (int k = 0; k < 12; k++) Jo[k] *= sc[k % 6];
#pragma unroll
            for (int k = 0; k < 6; k++) Jo[12 + k] *= sp[k % 3];
        }
        __syncthreads();   // rows of duplicate observations are read by other threads below
    }
    PSTAMP(6);
    // ---- phase 2: K-columns of cameras that do not see a point stay zero (Yt / Wt are cleared once per solve)
    for (int e = eb0 + tid; e < eb1; e += BM_T) {
        const int4 rec = A.erec[e];   // (observation, camera, point, dup flag) in one load
        const int flag = rec.w;
        if (flag == 2) continue;   // a later observation of the same (point, camera): folded into the first one
        const int i = rec.x, c = rec.y, pp = rec.z;
        const double* Jr = Jb + (size_t)i * 18;
        double jc[12], jp[6];
#pragma unroll
        for (int k = 0; k < 12; k++) jc[k] = Jr[k];
#pragma unroll
        for (int k = 0; k < 6; k++) jp[k] = Jr[12 + k];
        double w[18];
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 3; b++) w[a * 3 + b] = jc[a] * jp[b] + jc[6 + a] * jp[3 + b];
        }
        if (flag == 1) {
            for (int e2 = e + 1; e2 < A.pobs_start[pp + 1]; e2++) {
                const int i2 = A.pobs_list[e2];
                if (A.cam_idx[i2] != c) continue;
                const double* J2 = Jb + (size_t)i2 * 18;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int b = 0; b < 3; b++) w[a * 3 + b] += J2[a] * J2[12 + b] + J2[6 + a] * J2[15 + b];
                }
            }
        }
        const double* Ei = sEi + (pp - p0) * 9;
        double ei[9];
#pragma unroll
        for (int k = 0; k < 9; k++) ei[k] = Ei[k];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            double* wp = &A.Wd[(size_t)(6 * c + a) * krows + 3 * pp];
            double* yp = &A.Yd[(size_t)(6 * c + a) * krows + 3 * pp];
#pragma unroll
            for (int b = 0; b < 3; b++) {
                wp[b] = w[a * 3 + b];
                yp[b] = w[a * 3] * ei[0 * 3 + b] + w[a * 3 + 1] * ei[1 * 3 + b] + w[a * 3 + 2] * ei[2 * 3 + b];
            }
        }
    }
    PSTAMP(7);
#undef PSTAMP
}

constexpr int BM_WORK = BM_NW * 4 * 64;   // doubles of block-shared scratch: MFMA accumulators per wavefront / E^-1 of a block's points
// C|P kernel. From the second iteration on every block first takes the accept/reject decision on the previous step itself
// (same inputs, same arithmetic in every block; block 0 publishes the new state) — the state is double-buffered so nobody
// reads what block 0 writes. Blocks [0, cam_blocks): C role; the rest: P role.
__device__ inline void bam_campoint_role(const BAArgs& A, const BAGState* __restrict__ st_in, BAGState* __restrict__ st_out, bool decide,
                                         const double* __restrict__ part4, int nbp, int* __restrict__ chol_flag, int cam_blocks,
                                         double* Ublk, double* rhsblk, double* part_gmax, int bid, double* sred /* [BM_WORK] */) {
    __shared__ BAGState ss;
    __shared__ double ssc[8];
    static_assert(BM_PB * 12 <= BM_WORK, "LDS");
    if (threadIdx.x == 0) {
        ss = *st_in;
        if (decide) {
            if (!ss.done) bam_decide(ss, part4, nbp);
            if (bid == 0) *st_out = ss;
        }
    }
    __syncthreads();
    if (bid < cam_blocks) bam_cam_role(A, ss, bid, Ublk, rhsblk, sred, ssc);
    else bam_point_role(A, ss, chol_flag, bid - cam_blocks, part_gmax, sred);
}
__global__ __launch_bounds__(BM_T) void k_bam_campoint(BAArgs A, const BAGState* st_in, BAGState* st_out, int decide, const double* part4,
                                                       int nbp, int* chol_flag, int cam_blocks, double* Ublk, double* rhsblk,
                                                       double* part_gmax) { BACKEND_PRIO();
    __shared__ double swork[BM_WORK];
    bam_campoint_role(A, st_in, st_out, decide != 0, part4, nbp, chol_flag, cam_blocks, Ublk, rhsblk, part_gmax, blockIdx.x, swork);
}

// G: Schur contraction  G = Yt^T [Wt | g]  on FP64 MFMA: BG_H blocks per 16x16 tile, one wavefront per K-slice, the slices
// of a block are summed in slice order through LDS; the solve kernel adds the BG_H partials (fixed summation order).
constexpr int BG_W = 4, BG_H = 2;
__device__ inline void bam_gemm_role(const BAArgs& A, const BAGState* st, int bid, double* sacc /* [BG_W * 4 * 64] */) {
    static_assert(BG_W * 4 * 64 <= BM_NW * 4 * 64, "LDS");
    if (st->done) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, krows = A.krows, ldw = A.ldw;
    const int tc = A.tiles_c;
    const int tile = bid / BG_H, half = bid % BG_H;
    const int ti = tile / tc, tj = tile - ti * tc;
    const int kper = A.kper;   // multiple of 16; BG_H * BG_W * kper >= krows
    const int k0 = min(krows, (half * BG_W + wv) * kper), k1 = min(krows, k0 + kper);
    v4d acc = {0, 0, 0, 0};
    const double* pa = A.Yd + (size_t)(ti * 16 + (lane & 15)) * krows + 4 * (lane >> 4) + k0;
    const double* pb = A.Wd + (size_t)(tj * 16 + (lane & 15)) * krows + 4 * (lane >> 4) + k0;
    const int nk = __builtin_amdgcn_readfirstlane((k1 - k0) / 16);
    int kb = 0;
    for (; kb + 4 <= nk; kb += 4) {
        v4d a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a4[u] = *(const v4d*)(pa + 16 * u); b4[u] = *(const v4d*)(pb + 16 * u); }
        pa += 64; pb += 64;
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[u][q], b4[u][q], acc, 0, 0, 0);
        }
    }
    for (; kb < nk; kb++) {
        const v4d a4 = *(const v4d*)pa, b4 = *(const v4d*)pb;
        pa += 16; pb += 16;
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[q], b4[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) sacc[(wv * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double g = sacc[r * 64 + lane];
#pragma unroll
            for (int w = 1; w < BG_W; w++) g += sacc[(w * 4 + r) * 64 + lane];
            A.Gpart[((size_t)half * A.gp_rows + ti * 16 + (lane >> 4) + 4 * r) * ldw + tj * 16 + (lane & 15)] = g;
        }
    }
}

__global__ __launch_bounds__(64 * BG_W) void k_bam_gemm(BAArgs A, const BAGState* st) { BACKEND_PRIO();
    __shared__ double swork[BG_W * 4 * 64];
    bam_gemm_role(A, st, blockIdx.x, swork);
}

// S: loop-top tests, reduced camera system [S | rhs row] in LDS, right-looking Cholesky with the forward substitution
// folded in as row m (two barriers per column), backward substitution in registers of wavefront 0 -> step_c; rotation
// constants of the candidate cameras.
__device__ inline void bam_solve_role(const BAArgs& A, BAGState* st, const int* __restrict__ chol_flag, const double* __restrict__ part_cost,
                                      int nbo, const double* __restrict__ part_gmax, int nbp, const double* __restrict__ Ublk,
                                      const double* __restrict__ rhsblk, double* __restrict__ candrot) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ double red[3 * BM_NW];
    __shared__ BAGState ss;   // LDS copy of the state: read here, written through to global memory by thread 0
    __shared__ int sfail;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tx = tid & 15, ty = tid >> 4;
    unsigned long long t_prev = __builtin_readcyclecounter();
#define SSTAMP(k) do { if (A.stamps && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); A.stamps[k] += t_ - t_prev; t_prev = t_; } } while (0)
    if (tid == 0) { ss = *st; sfail = 0; }
    __syncthreads();
    if (ss.done) return;
    const int nc = A.nc, m = 6 * nc, n = m + 3 * A.np;
    double* S = dyn; double* dg = dyn + (size_t)(m + 1) * m;   // S: (m+1) x m, row m = right-hand side; dg: 1 / diagonal of L
    const double* x = A.x + (size_t)ss.cur * n;
    if (ss.need_eval) {   // block-uniform
        double g = 0, cs = 0;
        for (int i = tid; i < m; i += BM_T) g = fmax(g, fabs(rhsblk[i] / A.scale[i]));
        for (int i = tid; i < nbp; i += BM_T) g = fmax(g, part_gmax[i]);
        for (int i = tid; i < nbo; i += BM_T) cs += part_cost[i];
        cs = wave_sum_f64(cs);
        g = wave_max_f64(g);
        if (lane == 0) { red[BM_NW + wid] = cs; red[2 * BM_NW + wid] = g; }
        __syncthreads();
        if (tid == 0) {
            if (ss.first) {   // cost of the starting point (E0); later x_cost is the accepted candidate's cost
                const double c = (red[BM_NW] + red[BM_NW + 1]) + (red[BM_NW + 2] + red[BM_NW + 3]);
                ss.x_cost = st->x_cost = c;
                ss.initial_cost = st->initial_cost = c;
            }
            ss.gmax = st->gmax = fmax(fmax(red[2 * BM_NW], red[2 * BM_NW + 1]), fmax(red[2 * BM_NW + 2], red[2 * BM_NW + 3]));
        }
    }
    if (tid == 0) {
        int term = -1;
        if (ss.iter >= ss.max_iterations) term = 0;
        else if (ss.gmax >= 0 && ss.gmax <= 1e-10) term = 2;
        else if (ss.radius < 1e-32) term = 4;
        if (term >= 0) { ss.done = st->done = 1; ss.termination = st->termination = term; }
        st->step_valid = 0;
    }
    __syncthreads();
    if (ss.done) return;
    SSTAMP(16);
    // ---- S = blockdiag(U_c) + D2_c - Yt^T Wt (lower triangle), row m = rhs_c - Yt^T g
    {
        const double radius = ss.radius;
        const bool reuse = ss.reuse_diag != 0;
        for (int a = ty; a <= m; a += 16) {
            for (int b = tx; b < m && b <= a; b += 16) {
                double v;
                if (a < m) v = (a / 6 == b / 6) ? Ublk[(a / 6) * 36 + (a % 6) * 6 + (b % 6)] : 0.0;
                else v = rhsblk[b];
                // row m (the right-hand side) takes column m of G: (Yt^T g)[b]
                const size_t gi = (a < m) ? (size_t)a * A.ldw + b : (size_t)b * A.ldw + m;
                double g = A.Gpart[gi];
#pragma unroll
                for (int h = 1; h < BG_H; h++) g += A.Gpart[(size_t)h * A.gp_rows * A.ldw + gi];
                if (a == b) {
                    double d2;
                    if (!reuse) { d2 = fmin(fmax(v, 1e-6), 1e32); A.diag[a] = d2; }
                    else d2 = A.diag[a];
                    v += d2 / radius;
                }
                S[(size_t)a * m + b] = v - g;
            }
        }
    }
    __syncthreads();
    if (*chol_flag) return;   // a point block was not positive definite: invalid step (step_valid stays 0)
    SSTAMP(17);
    // Blocked right-looking Cholesky, block = one camera (6 columns). Per block: (a) one thread factors the 6x6 diagonal
    // block in registers (the only serial chain: 6 x sqrt + reciprocal), (b) one thread per row below solves its 6 entries
    // against the block (multiplications by the pivot reciprocals), (c) rank-6 update of the trailing lower triangle.
    // Row m carries the right-hand side, so the forward substitution is part of (b)/(c). dg keeps the pivot reciprocals.
    for (int jb = 0; jb < nc; jb++) {
        const int j0 = 6 * jb;
        if (tid == 0) {
            double a[21];   // lower triangle, row-major: (i,k) -> i*(i+1)/2 + k
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int k = 0; k <= i; k++) a[i * (i + 1) / 2 + k] = S[(size_t)(j0 + i) * m + j0 + k];
            }
            bool ok = true;
            double r[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const double d = a[j * (j + 1) / 2 + j];
                ok = ok && (d > 0.0);
                r[j] = rsqrt_nr(d);                      // pivot reciprocal; the diagonal of L itself is only kept for reference
                a[j * (j + 1) / 2 + j] = d * r[j];
#pragma unroll
                for (int i = j + 1; i < 6; i++) a[i * (i + 1) / 2 + j] *= r[j];
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
#pragma unroll
                    for (int k = j + 1; k <= i; k++) a[i * (i + 1) / 2 + k] -= a[i * (i + 1) / 2 + j] * a[k * (k + 1) / 2 + j];
                }
            }
#pragma unroll
            for (int i = 0; i < 6; i++) {
#pragma unroll
                for (int k = 0; k <= i; k++) S[(size_t)(j0 + i) * m + j0 + k] = a[i * (i + 1) / 2 + k];
                dg[j0 + i] = r[i];
            }
            if (!ok) sfail = 1;
        }
        __syncthreads();
        if (sfail) break;   // block-uniform
        {
            const int i = j0 + 6 + tid;   // rows below the block, row m = right-hand side (at most 6*22 - 6 + 1 < BM_T rows)
            if (i <= m) {
                double v[6], Lb[15], r[6];
#pragma unroll
                for (int k = 0; k < 6; k++) { v[k] = S[(size_t)i * m + j0 + k]; r[k] = dg[j0 + k]; }
#pragma unroll
                for (int q = 1; q < 6; q++) {
#pragma unroll
                    for (int k = 0; k < q; k++) Lb[q * (q - 1) / 2 + k] = S[(size_t)(j0 + q) * m + j0 + k];
                }
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    double t = v[q];
#pragma unroll
                    for (int k = 0; k < q; k++) t -= v[k] * Lb[q * (q - 1) / 2 + k];
                    v[q] = t * r[q];
                }
#pragma unroll
                for (int k = 0; k < 6; k++) S[(size_t)i * m + j0 + k] = v[k];
            }
        }
        __syncthreads();
        {
            const int r0 = j0 + 6;                 // first trailing row / column
            const int nr = m - r0;                 // trailing columns; rows r0..m (nr + 1 rows)
            const int total = nr * (nr + 1) / 2 + nr;   // lower triangle of the nr x nr block + the right-hand-side row
            for (int e = tid; e < total; e += BM_T) {
                int ii = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);   // row of the triangular index, then exact fix-up
                while ((ii + 1) * (ii + 2) / 2 <= e) ii++;
                while (ii * (ii + 1) / 2 > e) ii--;
                int kk = e - ii * (ii + 1) / 2;
                if (ii >= nr) { ii = nr; kk = e - nr * (nr + 1) / 2; }   // the last nr entries are the right-hand-side row
                const double* Li = S + (size_t)(r0 + ii) * m + j0;
                const double* Lk = S + (size_t)(r0 + kk) * m + j0;
                double acc = S[(size_t)(r0 + ii) * m + r0 + kk];
#pragma unroll
                for (int t = 0; t < 6; t++) acc -= Li[t] * Lk[t];
                S[(size_t)(r0 + ii) * m + r0 + kk] = acc;
            }
        }
        __syncthreads();
    }
    if (sfail) return;
    SSTAMP(18);
    if (wid == 0) {
        // L^T x = y by camera blocks from the last one up, in wavefront 0 (LDS ordering only, no block barrier): every lane
        // solves the 6x6 triangular block redundantly in registers, then the lanes subtract the block's contribution from the
        // rows above. Subtraction order per row = descending column index, as in the column-oriented scalar algorithm.
        double* ys = S + (size_t)m * m;   // row m: y on entry, x on exit
        for (int jb = nc - 1; jb >= 0; jb--) {
            const int j0 = 6 * jb;
            double xb[6], Lb[15], r[6];
#pragma unroll
            for (int q = 0; q < 6; q++) { xb[q] = ys[j0 + q]; r[q] = dg[j0 + q]; }
#pragma unroll
            for (int q = 1; q < 6; q++) {
#pragma unroll
                for (int k = 0; k < q; k++) Lb[q * (q - 1) / 2 + k] = S[(size_t)(j0 + q) * m + j0 + k];
            }
#pragma unroll
            for (int q = 5; q >= 0; q--) {
                double t = xb[q];
#pragma unroll
                for (int k = 5; k > q; k--) t -= Lb[k * (k - 1) / 2 + q] * xb[k];
                xb[q] = t * r[q];
            }
            for (int k = lane; k < j0; k += 64) {
                double acc = ys[k];
#pragma unroll
                for (int t = 5; t >= 0; t--) acc -= S[(size_t)(j0 + t) * m + k] * xb[t];
                ys[k] = acc;
            }
            if (lane < 6) {
                double v = xb[0];
#pragma unroll
                for (int t = 1; t < 6; t++) v = (lane == t) ? xb[t] : v;
                ys[j0 + lane] = v;
            }
            WAVE_SYNC();
        }
        for (int k = lane; k < m; k += 64) { const double v = -ys[k]; A.step[k] = v; S[k] = v; }   // the LM step is the negated solution
        WAVE_SYNC();
        SSTAMP(19);
        if (lane < nc) {   // candidate cameras and their rotation constants, once per camera instead of once per observation
            double cx[6];
#pragma unroll
            for (int k = 0; k < 6; k++) cx[k] = x[6 * lane + k] + S[6 * lane + k] * A.scale[6 * lane + k];
            CamRot cr;
            cam_rot_setup(cx, cr);
            double* o = candrot + 16 * lane;
#pragma unroll
            for (int k = 0; k < 6; k++) o[k] = cx[k];
            o[6] = cr.ct; o[7] = cr.st; o[8] = cr.ti; o[9] = cr.w0; o[10] = cr.w1; o[11] = cr.w2; o[12] = (double)cr.big;
        }
        if (lane == 0) st->step_valid = 1;
        SSTAMP(20);
    }
#undef SSTAMP
}

__global__ __launch_bounds__(BM_T) void k_bam_solve(BAArgs A, BAGState* st, const int* chol_flag, const double* part_cost, int nbo,
                                                    const double* part_gmax, int nbp, const double* Ublk, const double* rhsblk,
                                                    double* candrot) { BACKEND_PRIO();
    bam_solve_role(A, st, chol_flag, part_cost, nbo, part_gmax, nbp, Ublk, rhsblk, candrot);
}

// B: one block per BM_PB points. Phase A, per observation: Jp^T (Jc y_c); phase B, per point: back-substitution and the
// candidate point; phase C, per observation: model-cost-change and candidate-cost terms. Partials per block, fixed order.
__device__ inline void bam_backsub_role(const BAArgs& A, const BAGState* st, const double* __restrict__ candrot,
                                        double* __restrict__ tmp3, double* __restrict__ part4, int bid) {
    __shared__ double red[4 * BM_NW];
    __shared__ double sP[BM_PB * 6];   // per point of the block: step (3), candidate point (3)
    if (st->done || !st->step_valid) return;
    const int tid = threadIdx.x;
    unsigned long long t_prev = __builtin_readcyclecounter();
#define BSTAMP(k) do { if (A.stamps && bid == 0 && tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); A.stamps[k] += t_ - t_prev; t_prev = t_; } } while (0)
    const int nc = A.nc, m = 6 * nc, n = m + 3 * A.np;
    const int cur = st->cur;
    const double* x = A.x + (size_t)cur * n;
    double* cand = A.x + (size_t)(cur ^ 1) * n;
    const double* Jb = A.J + (size_t)cur * A.nobs * 18;          // r, J at the current point
    const double* resb = A.res + (size_t)cur * A.nobs * 2;
    double* Jn = A.J + (size_t)(cur ^ 1) * A.nobs * 18;          // r, J at the candidate (used if the step is accepted)
    double* resn = A.res + (size_t)(cur ^ 1) * A.nobs * 2;
    const bool spec = st->iter + 1 < st->max_iterations;         // after the last iteration nobody reads them
    const int p0 = bid * BM_PB, p1 = min(A.np, p0 + BM_PB);
    const int eb0 = A.pobs_start[p0], eb1 = A.pobs_start[p1];
    double mc = 0, cc = 0, dn2 = 0, xn2 = 0;
    for (int e = eb0 + tid; e < eb1; e += BM_T) {
        const int4 rec = A.erec[e];
        const int i = rec.x, c = rec.y;
        const double* Jr = Jb + (size_t)i * 18;
        double jy0 = 0, jy1 = 0;   // y_c = -step_c
#pragma unroll
        for (int a = 0; a < 6; a++) { const double yc = -A.step[6 * c + a]; jy0 += Jr[a] * yc; jy1 += Jr[6 + a] * yc; }
#pragma unroll
        for (int a = 0; a < 3; a++) tmp3[(size_t)e * 3 + a] = Jr[12 + a] * jy0 + Jr[15 + a] * jy1;
    }
    __syncthreads();
    BSTAMP(10);
    const int p = p0 + tid;
    if (tid < BM_PB && p < p1) {
        double t3[3] = {A.gp[(size_t)p * 3], A.gp[(size_t)p * 3 + 1], A.gp[(size_t)p * 3 + 2]};
        for (int e = A.pobs_start[p]; e < A.pobs_start[p + 1]; e++) {
#pragma unroll
            for (int a = 0; a < 3; a++) t3[a] -= tmp3[(size_t)e * 3 + a];
        }
        const double* Ei = A.Einv + (size_t)p * 9;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double sp = -(Ei[a * 3] * t3[0] + Ei[a * 3 + 1] * t3[1] + Ei[a * 3 + 2] * t3[2]);
            A.step[m + 3 * p + a] = sp;
            const double d = sp * A.scale[m + 3 * p + a];
            const double xo = x[m + 3 * p + a];
            xn2 += xo * xo;
            const double xp = xo + d;
            cand[m + 3 * p + a] = xp;
            sP[tid * 6 + a] = sp; sP[tid * 6 + 3 + a] = xp;
            dn2 += d * d;
        }
    }
    __syncthreads();
    BSTAMP(11);
    for (int e = eb0 + tid; e < eb1; e += BM_T) {
        const int4 rec = A.erec[e];
        const int i = rec.x, c = rec.y, pl = rec.z - p0;
        const double* Jr = Jb + (size_t)i * 18;
        double sc[6], sp[3], xp[3];
#pragma unroll
        for (int k = 0; k < 6; k++) sc[k] = A.step[6 * c + k];
#pragma unroll
        for (int k = 0; k < 3; k++) { sp[k] = sP[pl * 6 + k]; xp[k] = sP[pl * 6 + 3 + k]; }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            double mr = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) mr += Jr[rr * 6 + k] * sc[k];
#pragma unroll
            for (int k = 0; k < 3; k++) mr += Jr[12 + rr * 3 + k] * sp[k];
            mc -= mr * (resb[2 * i + rr] + mr / 2.0);
        }
        const double* o = candrot + 16 * c;
        CamRot cr;
        cr.ct = o[6]; cr.st = o[7]; cr.ti = o[8]; cr.w0 = o[9]; cr.w1 = o[10]; cr.w2 = o[11]; cr.big = (int)o[12];
        double r[2], rho0, rho1;
        if (spec) {
            double Jc[12], Jp[6];
            projection_residual_pre(cr, o, xp, A.obs[2 * i], A.obs[2 * i + 1], A.K, r, Jc, Jp, true);
            huber_rho(r[0] * r[0] + r[1] * r[1], A.huber, rho0, rho1);
            const double sr = sqrt(rho1);
            resn[2 * i] = r[0] * sr; resn[2 * i + 1] = r[1] * sr;
            const double* scc = A.scale + 6 * c;
            const double* scp = A.scale + 6 * A.nc + 3 * (pl + p0);
            double* Jo = Jn + (size_t)i * 18;
#pragma unroll
            for (int k = 0; k < 12; k++) Jo[k] = Jc[k] * sr * scc[k % 6];
#pragma unroll
            for (int k = 0; k < 6; k++) Jo[12 + k] = Jp[k] * sr * scp[k % 3];
        } else {
            projection_residual_pre(cr, o, xp, A.obs[2 * i], A.obs[2 * i + 1], A.K, r, nullptr, nullptr, false);
            huber_rho(r[0] * r[0] + r[1] * r[1], A.huber, rho0, rho1);
        }
        cc += 0.5 * rho0;
    }
    BSTAMP(12);
    if (bid == 0) {   // camera part of the candidate and of the step norm
        for (int i = tid; i < m; i += BM_T) {
            const double d = A.step[i] * A.scale[i];
            cand[i] = x[i] + d;
            dn2 += d * d;
            xn2 += x[i] * x[i];
        }
    }
    mc = wave_sum_f64(mc); cc = wave_sum_f64(cc); dn2 = wave_sum_f64(dn2); xn2 = wave_sum_f64(xn2);
    if ((tid & 63) == 0) { red[tid >> 6] = mc; red[BM_NW + (tid >> 6)] = cc; red[2 * BM_NW + (tid >> 6)] = dn2; red[3 * BM_NW + (tid >> 6)] = xn2; }
    __syncthreads();
    if (tid == 0) {
        part4[bid * 4] = (red[0] + red[1]) + (red[2] + red[3]);
        part4[bid * 4 + 1] = (red[BM_NW] + red[BM_NW + 1]) + (red[BM_NW + 2] + red[BM_NW + 3]);
        part4[bid * 4 + 2] = (red[2 * BM_NW] + red[2 * BM_NW + 1]) + (red[2 * BM_NW + 2] + red[2 * BM_NW + 3]);
        part4[bid * 4 + 3] = (red[3 * BM_NW] + red[3 * BM_NW + 1]) + (red[3 * BM_NW + 2] + red[3 * BM_NW + 3]);
    }
    BSTAMP(13);
#undef BSTAMP
}

__global__ __launch_bounds__(BM_T) void k_bam_backsub(BAArgs A, const BAGState* st, const double* candrot, double* tmp3, double* part4) { BACKEND_PRIO();
    bam_backsub_role(A, st, candrot, tmp3, part4, blockIdx.x);
}

// F: decision on the last step, results
__device__ inline void bam_finish_role(const BAArgs& A, const BAGState* __restrict__ st_in, const double* __restrict__ part4, int nbp) {
    __shared__ BAGState ss;
    if (threadIdx.x == 0) {
        ss = *st_in;
        if (!ss.done) bam_decide(ss, part4, nbp);
    }
    __syncthreads();
    const int n = 6 * A.nc + 3 * A.np;
    const double* x = A.x + (size_t)ss.cur * n;
    double* ocams = A.out ? A.out + 8 : A.cams;
    double* opts = A.out ? A.out + 8 + 6 * A.nc : A.pts;
    double* osum = A.out ? A.out : A.summary;
    for (int i = threadIdx.x; i < 6 * A.nc; i += BM_T) ocams[i] = x[i];
    for (int i = threadIdx.x; i < 3 * A.np; i += BM_T) opts[i] = x[6 * A.nc + i];
    if (threadIdx.x == 0) {
        osum[0] = ss.initial_cost; osum[1] = ss.x_cost; osum[2] = ss.iter; osum[3] = ss.successful;
        osum[4] = ss.termination;
    }
    if (A.out && A.done_seq) {   // the result block is complete: publish this call's sequence number (see pmv_ba_solve)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store((unsigned*)(A.out + 7), A.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(BM_T) void k_bam_finish(BAArgs A, const BAGState* st_in, const double* part4, int nbp) { BACKEND_PRIO();
    bam_finish_role(A, st_in, part4, nbp);
}

hipError_t launch_ba_multi(hipStream_t s, const BAArgs& A, void* d_state, double* d_part) {
    BAGState* st2[2] = {(BAGState*)d_state, (BAGState*)((char*)d_state + 256)};   // double-buffered state
    int* chol_flags = (int*)((char*)d_state + 512);                                // one "point block not SPD" flag per iteration
    static_assert(sizeof(BAGState) <= 256, "BAGState");
    const int nbo = (A.nobs + BM_T - 1) / BM_T, nbp = (A.np + BM_PB - 1) / BM_PB, m = 6 * A.nc;
    double* part_cost = d_part;                 // nbo
    double* part_gmax = part_cost + nbo;        // nbp
    double* part4 = part_gmax + nbp;            // nbp * 4
    double* Ublk = part4 + 4 * nbp;             // nc * 36
    double* rhsblk = Ublk + 36 * A.nc;          // nc * 6
    double* candrot = rhsblk + 6 * A.nc;        // nc * 16
    double* tmp3 = candrot + 16 * A.nc;         // nobs * 3
    const size_t shm = ((size_t)(m + 1) * m + (size_t)m) * sizeof(double);
    if (A.max_iterations > BA_MAX_ITERATIONS) return hipErrorInvalidValue;
    ProfScope ps(K_BA_LM, s);
    const int tiles = A.tiles_r * A.tiles_c;
    // launches: E0 (+ the clear of Yt | Wt), then per iteration C|P (with the decision on the previous step) -> G -> S -> B, then F
    const int clear_blocks = (int)std::min<size_t>(64, ((size_t)A.krows * A.ldw + 4 * BM_T - 1) / (4 * BM_T));
    { ProfScope p_(K_BAM_EVAL0, s);
    hipLaunchKernelGGL(k_bam_eval0, dim3(nbo + clear_blocks), dim3(BM_T), 0, s, A, st2[1], chol_flags, part_cost, nbo); }
    for (int it = 0; it < A.max_iterations; it++) {
        BAGState* sin = st2[it & 1];
        BAGState* sc = st2[(it + 1) & 1];   // the state of this iteration
        int* cf = chol_flags + it;
        if (it == 0) {   // the point kernel needs the camera scales of the C kernel in the first iteration
            ProfScope p_(K_BAM_CAMPOINT, s);
            hipLaunchKernelGGL(k_bam_campoint, dim3(A.nc), dim3(BM_T), 0, s, A, sc, sc, 0, part4, nbp, cf, A.nc, Ublk, rhsblk, part_gmax);
            hipLaunchKernelGGL(k_bam_campoint, dim3(nbp), dim3(BM_T), 0, s, A, sc, sc, 0, part4, nbp, cf, 0, Ublk, rhsblk, part_gmax);
        } else {
            ProfScope p_(K_BAM_CAMPOINT, s);
            hipLaunchKernelGGL(k_bam_campoint, dim3(A.nc + nbp), dim3(BM_T), 0, s, A, sin, sc, 1, part4, nbp, cf, A.nc, Ublk, rhsblk, part_gmax);
        }
        { ProfScope p_(K_BAM_GEMM, s);
        hipLaunchKernelGGL(k_bam_gemm, dim3(tiles * BG_H), dim3(64 * BG_W), 0, s, A, sc); }
        { ProfScope p_(K_BAM_SOLVE, s);
        hipLaunchKernelGGL(k_bam_solve, dim3(1), dim3(BM_T), shm, s, A, sc, cf, part_cost, nbo, part_gmax, nbp, Ublk, rhsblk, candrot); }
        { ProfScope p_(K_BAM_BACKSUB, s);
        hipLaunchKernelGGL(k_bam_backsub, dim3(nbp), dim3(BM_T), 0, s, A, sc, candrot, tmp3, part4); }
    }
    { ProfScope p_(K_BAM_FINISH, s);
    hipLaunchKernelGGL(k_bam_finish, dim3(1), dim3(BM_T), 0, s, A, st2[A.max_iterations & 1], part4, nbp); }
    return hipGetLastError();
}

// ---- batched launch chain: blockIdx.y = problem; every block runs exactly the role code of the single-problem kernels with the
// same block index, so each problem's result is bit-identical to its own launch_ba_multi chain ---------------------------------
void ba_fill_prob(BAProb& P, const BAArgs& A, void* d_state, double* d_part) {
    P.A = A;
    P.st2[0] = d_state; P.st2[1] = (char*)d_state + 256;
    P.chol_flags = (int*)((char*)d_state + 512);
    P.nbo = (A.nobs + BM_T - 1) / BM_T; P.nbp = (A.np + BM_PB - 1) / BM_PB;
    P.part_cost = d_part; P.part_gmax = P.part_cost + P.nbo; P.part4 = P.part_gmax + P.nbp; P.Ublk = P.part4 + 4 * P.nbp;
    P.rhsblk = P.Ublk + 36 * A.nc; P.candrot = P.rhsblk + 6 * A.nc; P.tmp3 = P.candrot + 16 * A.nc;
    P.clear_blocks = (int)std::min<size_t>(64, ((size_t)A.krows * A.ldw + 4 * BM_T - 1) / (4 * BM_T));
    P.tiles = A.tiles_r * A.tiles_c;
}
__global__ __launch_bounds__(BM_T) void k_bamB_eval0(const BAProb* __restrict__ probs) { BACKEND_PRIO();
    const BAProb& P = probs[blockIdx.y];
    const BAArgs A = P.A;   // by value (uniform address -> scalar loads once); a reference would be re-read after every store: 29 -> 57 us for k_bamB_campoint
    const int bx = blockIdx.x;
    if (bx < P.nbo) { bam_eval0_role(A, (BAGState*)P.st2[1], P.chol_flags, P.part_cost, bx, P.nbo); return; }
    const int zb = bx - P.nbo, nzb = P.clear_blocks;
    if (zb >= nzb) return;
    const size_t n2 = (size_t)A.krows * A.ldw;
    double2* z = (double2*)A.Yd;
    for (size_t i = (size_t)zb * BM_T + threadIdx.x; i < n2; i += (size_t)nzb * BM_T) z[i] = double2{0.0, 0.0};
}
// part: 0 = cameras only, 1 = points only (first iteration: the point role needs the camera scales), 2 = both with the decision
__global__ __launch_bounds__(BM_T) void k_bamB_campoint(const BAProb* __restrict__ probs, int it, int part) { BACKEND_PRIO();
    __shared__ double swork[BM_WORK];
    const BAProb& P = probs[blockIdx.y];
    const BAArgs A = P.A;   // by value (uniform address -> scalar loads once); a reference would be re-read after every store: 29 -> 57 us for k_bamB_campoint
    BAGState* sin = (BAGState*)P.st2[it & 1];
    BAGState* sc = (BAGState*)P.st2[(it + 1) & 1];
    int* cf = P.chol_flags + it;
    const int bx = blockIdx.x;
    if (part == 0) { if (bx >= A.nc) return; bam_campoint_role(A, sc, sc, false, P.part4, P.nbp, cf, A.nc, P.Ublk, P.rhsblk, P.part_gmax, bx, swork); }
    else if (part == 1) { if (bx >= P.nbp) return; bam_campoint_role(A, sc, sc, false, P.part4, P.nbp, cf, 0, P.Ublk, P.rhsblk, P.part_gmax, bx, swork); }
    else { if (bx >= A.nc + P.nbp) return; bam_campoint_role(A, sin, sc, true, P.part4, P.nbp, cf, A.nc, P.Ublk, P.rhsblk, P.part_gmax, bx, swork); }
}
__global__ __launch_bounds__(64 * BG_W) void k_bamB_gemm(const BAProb* __restrict__ probs, int it) { BACKEND_PRIO();
    __shared__ double swork[BG_W * 4 * 64];
    const BAProb& P = probs[blockIdx.y];
    if ((int)blockIdx.x >= P.tiles * BG_H) return;
    const BAArgs A = P.A;   // by value (uniform address -> scalar loads once); a reference would be re-read after every store: 29 -> 57 us for k_bamB_campoint
    bam_gemm_role(A, (const BAGState*)P.st2[(it + 1) & 1], blockIdx.x, swork);
}
__global__ __launch_bounds__(BM_T) void k_bamB_solve(const BAProb* __restrict__ probs, int it) { BACKEND_PRIO();
    const BAProb& P = probs[blockIdx.x];
    const BAArgs A = P.A;   // by value (uniform address -> scalar loads once); a reference would be re-read after every store: 29 -> 57 us for k_bamB_campoint
    bam_solve_role(A, (BAGState*)P.st2[(it + 1) & 1], P.chol_flags + it, P.part_cost, P.nbo, P.part_gmax, P.nbp, P.Ublk, P.rhsblk, P.candrot);
}
__global__ __launch_bounds__(BM_T) void k_bamB_backsub(const BAProb* __restrict__ probs, int it) { BACKEND_PRIO();
    const BAProb& P = probs[blockIdx.y];
    if ((int)blockIdx.x >= P.nbp) return;
    const BAArgs A = P.A;   // by value (uniform address -> scalar loads once); a reference would be re-read after every store: 29 -> 57 us for k_bamB_campoint
    bam_backsub_role(A, (const BAGState*)P.st2[(it + 1) & 1], P.candrot, P.tmp3, P.part4, blockIdx.x);
}
__global__ __launch_bounds__(BM_T) void k_bamB_finish(const BAProb* __restrict__ probs, int max_it) { BACKEND_PRIO();
    const BAProb& P = probs[blockIdx.x];
    const BAArgs A = P.A;   // by value (uniform address -> scalar loads once); a reference would be re-read after every store: 29 -> 57 us for k_bamB_campoint
    bam_finish_role(A, (const BAGState*)P.st2[max_it & 1], P.part4, P.nbp);
}
hipError_t launch_ba_multi_batch(hipStream_t s, const BAProb* d_probs, const BABatchDims& D) {
    if (D.n_probs <= 0) return hipSuccess;
    if (!d_probs || D.max_iterations < 1 || D.max_iterations > BA_MAX_ITERATIONS) return hipErrorInvalidValue;
    const size_t shm = ((size_t)(D.max_m + 1) * D.max_m + (size_t)D.max_m) * sizeof(double);
    if (shm > 150 * 1024) return hipErrorInvalidValue;
    const unsigned np = (unsigned)D.n_probs;
    ProfScope ps(K_BA_LM, s);
    hipLaunchKernelGGL(k_bamB_eval0, dim3(D.max_eval_blocks, np), dim3(BM_T), 0, s, d_probs);
    for (int it = 0; it < D.max_iterations; it++) {
        if (it == 0) {
            hipLaunchKernelGGL(k_bamB_campoint, dim3(D.max_nc, np), dim3(BM_T), 0, s, d_probs, it, 0);
            hipLaunchKernelGGL(k_bamB_campoint, dim3(D.max_nbp, np), dim3(BM_T), 0, s, d_probs, it, 1);
        } else
            hipLaunchKernelGGL(k_bamB_campoint, dim3(D.max_nc + D.max_nbp, np), dim3(BM_T), 0, s, d_probs, it, 2);
        hipLaunchKernelGGL(k_bamB_gemm, dim3(D.max_tiles * BG_H, np), dim3(64 * BG_W), 0, s, d_probs, it);
        hipLaunchKernelGGL(k_bamB_solve, dim3(np), dim3(BM_T), shm, s, d_probs, it);
        hipLaunchKernelGGL(k_bamB_backsub, dim3(D.max_nbp, np), dim3(BM_T), 0, s, d_probs, it);
    }
    hipLaunchKernelGGL(k_bamB_finish, dim3(np), dim3(BM_T), 0, s, d_probs, D.max_iterations);
    return hipGetLastError();
}

hipError_t launch_ba_residuals(hipStream_t s, const double* cams, const double* pts, const double* obs, const int* cam_idx,
                               const int* pt_idx, int nobs, const double* K, double* out_r, double* out_J) {
    if (nobs <= 0) return hipSuccess;
    ProfScope ps(K_BA_RESID, s);
    hipLaunchKernelGGL(k_ba_residuals, dim3((nobs + 255) / 256), dim3(256), 0, s, cams, pts, obs, cam_idx, pt_idx, nobs, K, out_r, out_J);
    return hipGetLastError();
}
hipError_t launch_ba_lm_batch(hipStream_t s, const BAArgs* d_args, int n_probs, int max_m) {
    if (!d_args || n_probs < 1 || max_m < 6) return hipErrorInvalidValue;
    const size_t shm = ((size_t)(max_m + 1) * max_m + (size_t)max_m) * sizeof(double);
    ProfScope ps(K_BA_LM, s);
    hipLaunchKernelGGL(k_ba_lm_batch, dim3(n_probs), dim3(BA_T), shm, s, d_args);
    return hipGetLastError();
}
hipError_t launch_ba_lm(hipStream_t s, const BAArgs& A) {
    const int m = 6 * A.nc;
    const size_t shm = ((size_t)(m + 1) * m + (size_t)m) * sizeof(double);   // [S | rhs row] + camera step, LDS-resident
    ProfScope ps(K_BA_LM, s);
    hipLaunchKernelGGL(k_ba_lm, dim3(1), dim3(BA_T), shm, s, A);
    return hipGetLastError();
}

// LDS opt-in (150 KB reduced camera system) per device; see frontend_prepare_device()
hipError_t backend_prepare_device() {
    hipError_t e = hipFuncSetAttribute((const void*)k_bam_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_bamB_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_ba_lm_batch, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_ba_lm, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

}  // namespace pmv
