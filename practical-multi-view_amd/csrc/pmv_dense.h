// Small dense FP64 routines shared by the back-end kernels: device copies of the fixed-choice algorithms of oracle/orc_math.h
// and host/vo_math.h (same operation order; compiled with -ffp-contract=off, so the results are bit-identical to the host).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>

namespace pmv {

// ---- small dense routines (device copies of the fixed-choice algorithms documented in DESIGN.md) -------------------------
// cyclic Jacobi, ascending eigenvalues. N is a compile-time constant and every loop over matrix indices is fully unrolled so
// that A, V and w stay in registers (with a run-time n they live in scratch memory and every access is an L2 round trip).
template <int N>
__device__ inline void d_jacobi_eig(double* A, double* w, double* V) {
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) V[i * N + j] = (i == j) ? 1.0 : 0.0;
    double tol_abs = 0;
#pragma unroll
    for (int i = 0; i < N; i++) tol_abs += fabs(A[i * N + i]);
    tol_abs *= 1e-17;
    for (int sweep = 0; sweep < 30; sweep++) {
        int rotated = 0;
#pragma unroll
        for (int p = 0; p < N - 1; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (apq != 0.0) {
                    const double app = A[p * N + p], aqq = A[q * N + q];
                    if (fabs(apq) <= tol_abs) { A[p * N + q] = A[q * N + p] = 0.0; }
                    else {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            if (k == p || k == q) continue;
                            const double akp = A[k * N + p], akq = A[k * N + q];
                            const double nkp = c * akp - s * akq, nkq = s * akp + c * akq;
                            A[k * N + p] = A[p * N + k] = nkp;
                            A[k * N + q] = A[q * N + k] = nkq;
                        }
                        A[p * N + p] = app - t * apq;
                        A[q * N + q] = aqq + t * apq;
                        A[p * N + q] = A[q * N + p] = 0.0;
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            const double vkp = V[k * N + p], vkq = V[k * N + q];
                            V[k * N + p] = c * vkp - s * vkq;
                            V[k * N + q] = s * vkp + c * vkq;
                        }
                        rotated++;
                    }
                }
            }
        if (!rotated) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) w[i] = A[i * N + i];
    // ascending selection sort as a fixed compare-exchange network over (i, j > i): same result as "find min, swap"
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        // index of the minimum of w[i..N-1] (first occurrence), then swap columns i and m
        int m = i;
        double wm = w[i];
#pragma unroll
        for (int j = i + 1; j < N; j++) if (w[j] < wm) { wm = w[j]; m = j; }
#pragma unroll
        for (int j = i + 1; j < N; j++) {
            if (j == m) {
                double t = w[i]; w[i] = w[j]; w[j] = t;
#pragma unroll
                for (int k = 0; k < N; k++) { t = V[k * N + i]; V[k * N + i] = V[k * N + j]; V[k * N + j] = t; }
            }
        }
    }
}

__device__ inline void d_svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double AtA[9], w[3], Ve[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = acc;
        }
    d_jacobi_eig<3>(AtA, w, Ve);
    for (int k = 0; k < 3; k++) {
        const int src = 2 - k;
        s[k] = sqrt(w[src] > 0 ? w[src] : 0.0);
        for (int i = 0; i < 3; i++) V[i * 3 + k] = Ve[i * 3 + src];
    }
    for (int k = 0; k < 3; k++) {
        double u[3];
        for (int i = 0; i < 3; i++) u[i] = A[i * 3] * V[0 * 3 + k] + A[i * 3 + 1] * V[1 * 3 + k] + A[i * 3 + 2] * V[2 * 3 + k];
        const double nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (nrm > 1e-12 * (s[0] > 0 ? s[0] : 1.0) && nrm > 0) {
            for (int i = 0; i < 3; i++) U[i * 3 + k] = u[i] / nrm;
        } else if (k == 2) {
            U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
            U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
            U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
        } else {
            for (int i = 0; i < 3; i++) U[i * 3 + k] = (i == k) ? 1.0 : 0.0;
        }
    }
}

__device__ inline void d_rodrigues_v2m(const double r[3], double R[9]) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

__device__ inline void d_rodrigues_m2v(const double Rin[9], double r[3]) {
    double U[9], s[3], V[9], R[9];
    d_svd3(Rin, U, s, V);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double sn = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (sn < 1e-5) {
        if (c > 0) { rx = ry = rz = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = sqrt(fmax(t, 0.));
            t = (R[4] + 1) * 0.5; ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        const double vth = 1 / (2 * sn) * theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

template <int M, int N>
__device__ inline void d_pinv_solve(const double* A, const double* b, double* x) {
    double AtA[N * N], w[N], V[N * N], Atb[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < M; k++) acc += A[k * N + i] * A[k * N + j];
            AtA[i * N + j] = acc;
        }
        double acc = 0;
#pragma unroll
        for (int k = 0; k < M; k++) acc += A[k * N + i] * b[k];
        Atb[i] = acc;
    }
    d_jacobi_eig<N>(AtA, w, V);
    double ssum = 0;
#pragma unroll
    for (int i = 0; i < N; i++) ssum += sqrt(w[i] > 0 ? w[i] : 0.0);
    const double thr = 2 * DBL_EPSILON * ssum;
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double sg = sqrt(w[k] > 0 ? w[k] : 0.0);
        if (sg > thr) {
            double proj = 0;
#pragma unroll
            for (int i = 0; i < N; i++) proj += V[i * N + k] * Atb[i];
            proj /= w[k];
#pragma unroll
            for (int i = 0; i < N; i++) x[i] += V[i * N + k] * proj;
        }
    }
}

__device__ inline double d_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline double d_dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

}  // namespace pmv
