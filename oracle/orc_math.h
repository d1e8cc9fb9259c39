// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
// Small dense linear algebra used by the PnP / five-point / BA restatements: plain loops, double precision,
// no FMA contraction (build flags).  The eigen/SVD routines are this restatement's FIXED CHOICE where OpenCV
// uses its own JacobiSVD / LAPACK: classic cyclic Jacobi with a fixed (p<q, row-major) sweep order.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cfloat>

namespace orc {

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


// Symmetric eigen-decomposition with the PARALLEL (round-robin) Jacobi ordering, n even (used for EPnP's 12x12 M^T M).
// One round = n/2 disjoint rotations whose angles are all taken from the matrix at the start of the round, applied as
//   B = A J (columns), A' = J^T B (rows), V' = V J,
// every element update being the two-term expression written below. This is the FIXED eigen-solver of this restatement
// for the 12x12 case: a wavefront evaluates a round with one lane per element and obtains the same bits.
// Round r (0..n-2) pairs: (n-1, r) and ((r+k) mod (n-1), (r-k) mod (n-1)) for k = 1..n/2-1, each ordered p < q.
inline void jacobi_eig_parallel(double* A, int n, double* w, double* V, int max_sweeps = 30) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    std::vector<double> B((size_t)n * n);
    std::vector<int> P(n / 2), Q(n / 2);
    std::vector<double> C(n / 2), S(n / 2);
    double tol_abs = 0;
    for (int i = 0; i < n; i++) tol_abs += std::fabs(A[i * n + i]);
    tol_abs *= 1e-17;
    for (int sweep = 0; sweep < max_sweeps; sweep++) {
        int rotated = 0;
        for (int r = 0; r < n - 1; r++) {
            for (int g = 0; g < n / 2; g++) {
                int a, b;
                if (g == 0) { a = n - 1; b = r; }
                else { a = (r + g) % (n - 1); b = (r - g + (n - 1)) % (n - 1); }
                const int p = a < b ? a : b, q = a < b ? b : a;
                P[g] = p; Q[g] = q;
                const double apq = A[p * n + q], app = A[p * n + p], aqq = A[q * n + q];
                if (std::fabs(apq) <= tol_abs) { C[g] = 1.0; S[g] = 0.0; continue; }
                (void)app;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                C[g] = 1.0 / std::sqrt(t * t + 1.0);
                S[g] = t * C[g];
                rotated++;
            }
            // B = A J
            for (int i = 0; i < n; i++)
                for (int g = 0; g < n / 2; g++) {
                    const double x = A[i * n + P[g]], y = A[i * n + Q[g]];
                    B[i * n + P[g]] = C[g] * x - S[g] * y;
                    B[i * n + Q[g]] = S[g] * x + C[g] * y;
                }
            // A = J^T B
            for (int j = 0; j < n; j++)
                for (int g = 0; g < n / 2; g++) {
                    const double x = B[P[g] * n + j], y = B[Q[g] * n + j];
                    A[P[g] * n + j] = C[g] * x - S[g] * y;
                    A[Q[g] * n + j] = S[g] * x + C[g] * y;
                }
            // V = V J
            for (int i = 0; i < n; i++)
                for (int g = 0; g < n / 2; g++) {
                    const double x = V[i * n + P[g]], y = V[i * n + Q[g]];
                    V[i * n + P[g]] = C[g] * x - S[g] * y;
                    V[i * n + Q[g]] = S[g] * x + C[g] * y;
                }
        }
        if (!rotated) break;
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) {
        int mi = i;
        for (int j = i + 1; j < n; j++) if (w[j] < w[mi]) mi = j;
        if (mi != i) {
            std::swap(w[i], w[mi]);
            for (int k = 0; k < n; k++) std::swap(V[k * n + i], V[k * n + mi]);
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


// ceres::AngleAxisRotatePoint (rotation.h) plus its exact derivatives: p = R(a) q,
// dpdw = d p / d a (row-major 3x3), Rm = d p / d q.  Two branches exactly as Ceres evaluates them.
inline void angle_axis_rotate(const double a[3], const double q[3], double p[3], double dpdw[9], double Rm[9]) {
    const double theta2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = std::sqrt(theta2);
        const double ct = std::cos(theta), st = std::sin(theta), ti = 1.0 / theta;
        const double w[3] = {a[0] * ti, a[1] * ti, a[2] * ti};
        const double wxq[3] = {w[1] * q[2] - w[2] * q[1], w[2] * q[0] - w[0] * q[2], w[0] * q[1] - w[1] * q[0]};
        const double wq = w[0] * q[0] + w[1] * q[1] + w[2] * q[2];
        const double tmp = wq * (1.0 - ct);
        for (int i = 0; i < 3; i++) p[i] = q[i] * ct + wxq[i] * st + w[i] * tmp;
        const double c1 = 1.0 - ct;
        Rm[0] = ct + c1 * w[0] * w[0];        Rm[1] = c1 * w[0] * w[1] - st * w[2]; Rm[2] = c1 * w[0] * w[2] + st * w[1];
        Rm[3] = c1 * w[1] * w[0] + st * w[2]; Rm[4] = ct + c1 * w[1] * w[1];        Rm[5] = c1 * w[1] * w[2] - st * w[0];
        Rm[6] = c1 * w[2] * w[0] - st * w[1]; Rm[7] = c1 * w[2] * w[1] + st * w[0]; Rm[8] = ct + c1 * w[2] * w[2];
        double dw[9];   // dw/da = (I - w w^T)/theta ; dtheta/da = w
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) dw[i * 3 + k] = ((i == k ? 1.0 : 0.0) - w[i] * w[k]) * ti;
        for (int k = 0; k < 3; k++) {
            const double dwk[3] = {dw[0 * 3 + k], dw[1 * 3 + k], dw[2 * 3 + k]};
            const double dwxq[3] = {dwk[1] * q[2] - dwk[2] * q[1], dwk[2] * q[0] - dwk[0] * q[2], dwk[0] * q[1] - dwk[1] * q[0]};
            const double dwq = dwk[0] * q[0] + dwk[1] * q[1] + dwk[2] * q[2];
            const double dct = -st * w[k], dst = ct * w[k];
            const double dtmp = dwq * (1.0 - ct) + wq * (st * w[k]);
            for (int i = 0; i < 3; i++)
                dpdw[i * 3 + k] = q[i] * dct + dwxq[i] * st + wxq[i] * dst + dwk[i] * tmp + w[i] * dtmp;
        }
    } else {
        const double wxq[3] = {a[1] * q[2] - a[2] * q[1], a[2] * q[0] - a[0] * q[2], a[0] * q[1] - a[1] * q[0]};
        for (int i = 0; i < 3; i++) p[i] = q[i] + wxq[i];
        dpdw[0] = 0;     dpdw[1] = q[2];  dpdw[2] = -q[1];
        dpdw[3] = -q[2]; dpdw[4] = 0;     dpdw[5] = q[0];
        dpdw[6] = q[1];  dpdw[7] = -q[0]; dpdw[8] = 0;
        Rm[0] = 1;     Rm[1] = -a[2]; Rm[2] = a[1];
        Rm[3] = a[2];  Rm[4] = 1;     Rm[5] = -a[0];
        Rm[6] = -a[1]; Rm[7] = a[0];  Rm[8] = 1;
    }
}

// Least squares x = pinv(A) b for A (m x n, row-major, n <= 6) via eigen-decomposition of A^T A; singular values
// <= 2*DBL_EPSILON*sum(sigma) are treated as zero (the cv::SVD::backSubst rule behind cvSolve/cvInvert(CV_SVD)).
inline void pinv_solve(const double* A, int m, int n, const double* b, double* x) {
    double AtA[36], w[6], V[36], Atb[6];
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            double acc = 0;
            for (int k = 0; k < m; k++) acc += A[k * n + i] * A[k * n + j];
            AtA[i * n + j] = acc;
        }
        double acc = 0;
        for (int k = 0; k < m; k++) acc += A[k * n + i] * b[k];
        Atb[i] = acc;
    }
    jacobi_eig(AtA, n, w, V);
    double ssum = 0;
    for (int i = 0; i < n; i++) ssum += std::sqrt(w[i] > 0 ? w[i] : 0.0);
    const double thr = 2 * DBL_EPSILON * ssum;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int k = 0; k < n; k++) {
        const double sg = std::sqrt(w[k] > 0 ? w[k] : 0.0);
        if (!(sg > thr)) continue;
        double proj = 0;
        for (int i = 0; i < n; i++) proj += V[i * n + k] * Atb[i];
        proj /= w[k];
        for (int i = 0; i < n; i++) x[i] += V[i * n + k] * proj;
    }
}

// In-place Cholesky (lower) of an n x n SPD matrix (row-major). Returns false if a pivot is <= 0.
inline bool cholesky(double* A, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = A[i * n + j];
            for (int k = 0; k < j; k++) v -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = v / d;
        }
    }
    return true;
}
inline void cholesky_solve(const double* L, int n, double* b) {
    for (int i = 0; i < n; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= L[i * n + k] * b[k];
        b[i] = v / L[i * n + i];
    }
    // backward substitution, column-oriented (row k receives its updates in DESCENDING i): the order a lane-per-row
    // wavefront evaluates
    for (int i = n - 1; i >= 0; i--) {
        b[i] = b[i] / L[i * n + i];
        for (int k = 0; k < i; k++) b[k] -= L[i * n + k] * b[i];
    }
}

// Gaussian elimination with partial pivoting, A (n x n) and b (n) destroyed; returns false if singular.
inline bool gauss_solve(double* A, double* b, int n) {
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
        if (A[piv * n + c] == 0.0) return false;
        if (piv != c) {
            for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; r++) {
            const double f = A[r * n + c] / A[c * n + c];
            if (f == 0.0) continue;
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double v = b[r];
        for (int k = r + 1; k < n; k++) v -= A[r * n + k] * b[k];
        b[r] = v / A[r * n + r];
    }
    return true;
}

}  // namespace orc
