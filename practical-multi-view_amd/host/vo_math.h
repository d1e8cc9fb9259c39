// Small dense linear algebra for the HOST side of the product (adapters' Rodrigues conversions, five-point
// triangulator).  Plain double loops, cyclic Jacobi with a fixed sweep order; see DESIGN.md "fixed choices".
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cfloat>

namespace vmath {

// Symmetric eigen-decomposition, cyclic Jacobi. A (n x n, row-major) is destroyed; on return w[k] ascending and
// V[:,k] (column k, V row-major n x n) the matching unit eigenvector. Fixed 30 sweeps max, exits when off-diagonal
// mass is exactly below 1e-300 or no rotation was applied in a sweep.
inline void jacobi_eig(double* A, int n, double* w, double* V, int max_sweeps = 30) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    // off-diagonals below 1e-17 * trace are noise (they would only rotate rounding errors inside null spaces for ever)
    double tol_abs = 0;
    for (int i = 0; i < n; i++) tol_abs += std::fabs(A[i * n + i]);
    tol_abs *= 1e-17;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        int rotated = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double app = A[p * n + p], aqq = A[q * n + q];
                if (std::fabs(apq) <= tol_abs) { A[p * n + q] = A[q * n + p] = 0.0; continue; }
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    if (k == p || k == q) continue;
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    const double nkp = c * akp - s * akq, nkq = s * akp + c * akq;
                    A[k * n + p] = A[p * n + k] = nkp;
                    A[k * n + q] = A[q * n + k] = nkq;
                }
                A[p * n + p] = app - t * apq;
                A[q * n + q] = aqq + t * apq;
                A[p * n + q] = A[q * n + p] = 0.0;
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
                rotated++;
            }
        if (!rotated) break;
    }
    // ascending selection sort (stable w.r.t. original index on ties), permuting V's columns
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++) if (w[j] < w[m]) m = j;
        if (m != i) {
            std::swap(w[i], w[m]);
            for (int k = 0; k < n; k++) std::swap(V[k * n + i], V[k * n + m]);
        }
    }
}

// 3x3 SVD A = U diag(s) V^T via eigen-decomposition of A^T A (s descending). Rank-deficient columns of U are
// completed by cross products. A row-major.
inline void svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double AtA[9], w[3], Ve[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = acc;
        }
    jacobi_eig(AtA, 3, w, Ve);
    // descending order
    for (int k = 0; k < 3; k++) {
        const int src = 2 - k;
        s[k] = std::sqrt(w[src] > 0 ? w[src] : 0.0);
        for (int i = 0; i < 3; i++) V[i * 3 + k] = Ve[i * 3 + src];
    }
    for (int k = 0; k < 3; k++) {
        double u[3];
        for (int i = 0; i < 3; i++) u[i] = A[i * 3] * V[0 * 3 + k] + A[i * 3 + 1] * V[1 * 3 + k] + A[i * 3 + 2] * V[2 * 3 + k];
        const double nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
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

inline double det3(const double M[9]) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, t, sizeof(t));
}
inline void mat3_T(const double A[9], double B[9]) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[j * 3 + i];
    memcpy(B, t, sizeof(t));
}

// cv::Rodrigues vector -> matrix (calib3d cvRodrigues2)
inline void rodrigues_v2m(const double r[3], double R[9]) {
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    R[0] = c + c1 * x * x;     R[1] = c1 * x * y - s * z; R[2] = c1 * x * z + s * y;
    R[3] = c1 * x * y + s * z; R[4] = c + c1 * y * y;     R[5] = c1 * y * z - s * x;
    R[6] = c1 * x * z - s * y; R[7] = c1 * y * z + s * x; R[8] = c + c1 * z * z;
}

// cv::Rodrigues matrix -> vector (SVD-orthonormalised first, as cvRodrigues2 does)
inline void rodrigues_m2v(const double Rin[9], double r[3]) {
    double U[9], s[3], V[9], R[9], Vt[9];
    svd3(Rin, U, s, V);
    mat3_T(V, Vt);
    mat3_mul(U, Vt, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double sn = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (sn < 1e-5) {
        if (c > 0) { rx = ry = rz = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = std::sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5; ry = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        const double vth = 1 / (2 * sn) * theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}


}  // namespace vmath
