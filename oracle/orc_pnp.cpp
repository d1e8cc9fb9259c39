// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).  PARITY UNPINNED (OpenCV internals).
//
// Restates cv::solvePnPRansac(obj, img, K, noDist, rvec, tvec, useExtrinsicGuess=true, 100, 8.f, 0.99, inliers)
// as called at /root/reference/OpenCVEPnPSolver.cpp:35-36 (default flags), from the published OpenCV 3.4 algorithm
// (calib3d/solvepnp.cpp, ptsetreg.cpp, epnp.cpp):
//   RANSACPointSetRegistrator: RNG((uint64)-1) MWC generator, 5 distinct indices per sample, model = EPnP on the sample
//   (rvec via Rodrigues), error = squared float32 reprojection distance, inlier <= 8^2, best = most inliers (> max(prev,4)),
//   adaptive niters = RANSACUpdateNumIters(0.99, outlier ratio, 5, niters); then SOLVEPNP_ITERATIVE with
//   useExtrinsicGuess on the inliers (CvLevMarq: 6 params, <=20 iterations, eps FLT_EPSILON, lambda 10^-3 .. 10^16).
// FIXED CHOICES where 3.4.x point releases / builds differ:
//   * the LM refit starts from the EPnP model of the LAST RANSAC iteration executed (in 3.4 the callback writes every
//     model into the caller's rvec/tvec buffers, which the refit then reads as its guess — quirk Q8/Q9);
//   * SVD-based steps (PCA of the control points, 12x12 null space, 3x3 alignment, least squares) use the cyclic-Jacobi
//     routines of orc_math.h; the LM normal equations are solved by Gaussian elimination with partial pivoting
//     (OpenCV: DECOMP_SVD) — same solution for the non-singular 6x6 systems that occur.
#include "orc_api.h"
#include "orc_math.h"
#include <cstring>
#include <cfloat>

namespace orc {

struct RNG {   // cv::RNG
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline double dist2(const double* a, const double* b) {
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// EPnP (Lepetit/Moreno-Noguer/Fua) as shipped in OpenCV's epnp.cpp
struct EPnP {
    int n;
    double uc, vc, fu, fv;
    std::vector<double> pws, us, alphas, pcs;
    double cws[4][3], ccs[4][3];

    EPnP(const double K[9], const float* obj, const float* img, const int* idx, int n_) : n(n_) {
        fu = K[0]; fv = K[4]; uc = K[2]; vc = K[5];
        pws.resize(3 * n); us.resize(2 * n); alphas.resize(4 * n); pcs.resize(3 * n);
        const double ifx = 1. / fu, ify = 1. / fv;
        for (int i = 0; i < n; i++) {
            const int s = idx ? idx[i] : i;
            pws[3 * i] = obj[3 * s]; pws[3 * i + 1] = obj[3 * s + 1]; pws[3 * i + 2] = obj[3 * s + 2];
            // cv::undistortPoints (no distortion) -> float32 normalised point, then epnp::init_points
            const float xn = (float)(((double)img[2 * s] - uc) * ifx);
            const float yn = (float)(((double)img[2 * s + 1] - vc) * ify);
            us[2 * i] = xn * fu + uc;
            us[2 * i + 1] = yn * fv + vc;
        }
    }

    void choose_control_points() {
        cws[0][0] = cws[0][1] = cws[0][2] = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[0][j] /= n;
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double acc = 0;
                for (int i = 0; i < n; i++) acc += (pws[3 * i + a] - cws[0][a]) * (pws[3 * i + b] - cws[0][b]);
                C[a * 3 + b] = acc;
            }
        double w[3], V[9];
        jacobi_eig(C, 3, w, V);   // ascending; OpenCV's SVD is descending -> reverse
        for (int i = 1; i < 4; i++) {
            const int src = 3 - i;
            const double k = std::sqrt((w[src] > 0 ? w[src] : 0.0) / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * V[j * 3 + src];
        }
    }

    void compute_barycentric_coordinates() {
        double cc[9], U[9], s[3], V[9], ci[9];
        for (int i = 0; i < 3; i++)
            for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        svd3(cc, U, s, V);   // cvInvert(CV_SVD): pseudo-inverse, singular values <= 2 eps sum(s) dropped
        const double thr = 2 * DBL_EPSILON * (s[0] + s[1] + s[2]);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double acc = 0;
                for (int k = 0; k < 3; k++)
                    if (s[k] > thr) acc += V[i * 3 + k] * U[j * 3 + k] / s[k];
                ci[i * 3 + j] = acc;
            }
        for (int i = 0; i < n; i++) {
            const double* pi = &pws[3 * i];
            double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }

    void compute_ccs(const double* betas, const double* v4) {   // v4[i] = i-th smallest eigenvector (12)
        for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
        for (int i = 0; i < 4; i++) {
            const double* v = v4 + 12 * i;
            for (int j = 0; j < 4; j++)
                for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
        }
    }
    void compute_pcs() {
        for (int i = 0; i < n; i++) {
            const double* a = &alphas[4 * i];
            double* pc = &pcs[3 * i];
            for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
    }
    void solve_for_sign() {
        if (pcs[2] < 0.0) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
            for (int i = 0; i < n; i++) { pcs[3 * i] = -pcs[3 * i]; pcs[3 * i + 1] = -pcs[3 * i + 1]; pcs[3 * i + 2] = -pcs[3 * i + 2]; }
        }
    }
    void estimate_R_and_t(double R[3][3], double t[3]) {
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
        for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
        double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++) {
            const double* pc = &pcs[3 * i];
            const double* pw = &pws[3 * i];
            for (int j = 0; j < 3; j++) {
                abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
                abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
                abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
            }
        }
        double U[9], s[3], V[9];
        svd3(abt, U, s, V);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R[i][j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
        const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                           R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
        if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
        t[0] = pc0[0] - dot3(R[0], pw0);
        t[1] = pc0[1] - dot3(R[1], pw0);
        t[2] = pc0[2] - dot3(R[2], pw0);
    }
    double reprojection_error(const double R[3][3], const double t[3]) {
        double sum2 = 0.0;
        for (int i = 0; i < n; i++) {
            const double* pw = &pws[3 * i];
            const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1], inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
            const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
            const double u = us[2 * i], v = us[2 * i + 1];
            sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n;
    }
    double compute_R_and_t(const double* v4, const double* betas, double R[3][3], double t[3]) {
        compute_ccs(betas, v4);
        compute_pcs();
        solve_for_sign();
        estimate_R_and_t(R, t);
        return reprojection_error(R, t);
    }

    static void compute_L_6x10(const double* v4, double* l) {
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            const double* v = v4 + 12 * i;
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                dv[i][j][0] = v[3 * a] - v[3 * b];
                dv[i][j][1] = v[3 * a + 1] - v[3 * b + 1];
                dv[i][j][2] = v[3 * a + 2] - v[3 * b + 2];
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        for (int i = 0; i < 6; i++) {
            double* row = l + 10 * i;
            row[0] = dot3(dv[0][i], dv[0][i]);
            row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
            row[2] = dot3(dv[1][i], dv[1][i]);
            row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
            row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
            row[5] = dot3(dv[2][i], dv[2][i]);
            row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
            row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
            row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
            row[9] = dot3(dv[3][i], dv[3][i]);
        }
    }
    void compute_rho(double* rho) {
        rho[0] = dist2(cws[0], cws[1]); rho[1] = dist2(cws[0], cws[2]); rho[2] = dist2(cws[0], cws[3]);
        rho[3] = dist2(cws[1], cws[2]); rho[4] = dist2(cws[1], cws[3]); rho[5] = dist2(cws[2], cws[3]);
    }
    static void find_betas_approx_1(const double* L, const double* rho, double* betas) {
        double l4[24], b4[4];
        for (int i = 0; i < 6; i++) { l4[i * 4] = L[i * 10]; l4[i * 4 + 1] = L[i * 10 + 1]; l4[i * 4 + 2] = L[i * 10 + 3]; l4[i * 4 + 3] = L[i * 10 + 6]; }
        pinv_solve(l4, 6, 4, rho, b4);
        if (b4[0] < 0) { betas[0] = std::sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
        else { betas[0] = std::sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0]; }
    }
    static void find_betas_approx_2(const double* L, const double* rho, double* betas) {
        double l3[18], b3[3];
        for (int i = 0; i < 6; i++) { l3[i * 3] = L[i * 10]; l3[i * 3 + 1] = L[i * 10 + 1]; l3[i * 3 + 2] = L[i * 10 + 2]; }
        pinv_solve(l3, 6, 3, rho, b3);
        if (b3[0] < 0) { betas[0] = std::sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
        else { betas[0] = std::sqrt(b3[0]); betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) betas[0] = -betas[0];
        betas[2] = 0.0; betas[3] = 0.0;
    }
    static void find_betas_approx_3(const double* L, const double* rho, double* betas) {
        double l5[30], b5[5];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 5; j++) l5[i * 5 + j] = L[i * 10 + j];
        pinv_solve(l5, 6, 5, rho, b5);
        if (b5[0] < 0) { betas[0] = std::sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
        else { betas[0] = std::sqrt(b5[0]); betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) betas[0] = -betas[0];
        betas[2] = b5[3] / betas[0];
        betas[3] = 0.0;
    }
    // epnp::qr_solve (Householder QR of a 6x4 system), restated literally including its eta scan
    static void qr_solve(double* A, double* b, double* X) {
        const int nr = 6, nc = 4;
        double A1[6], A2[6];
        double* ppAkk = A;
        for (int k = 0; k < nc; k++) {
            double* ppAik1 = ppAkk;
            double eta = std::fabs(*ppAik1);
            for (int i = k + 1; i < nr; i++) {
                const double elt = std::fabs(*ppAik1);
                if (eta < elt) eta = elt;
                ppAik1 += nc;
            }
            if (eta == 0) { A1[k] = A2[k] = 0.0; X[0] = X[1] = X[2] = X[3] = 0.0; return; }
            double* ppAik2 = ppAkk;
            double sum2 = 0.0;
            const double inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) { *ppAik2 *= inv_eta; sum2 += *ppAik2 * *ppAik2; ppAik2 += nc; }
            double sigma = std::sqrt(sum2);
            if (*ppAkk < 0) sigma = -sigma;
            *ppAkk += sigma;
            A1[k] = sigma * *ppAkk;
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double* ppAik = ppAkk;
                double sum = 0;
                for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
                const double tau = sum / A1[k];
                ppAik = ppAkk;
                for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
            }
            ppAkk += nc + 1;
        }
        double* ppAjj = A;
        for (int j = 0; j < nc; j++) {
            double* ppAij = ppAjj;
            double tau = 0;
            for (int i = j; i < nr; i++) { tau += *ppAij * b[i]; ppAij += nc; }
            tau /= A1[j];
            ppAij = ppAjj;
            for (int i = j; i < nr; i++) { b[i] -= tau * *ppAij; ppAij += nc; }
            ppAjj += nc + 1;
        }
        X[nc - 1] = b[nc - 1] / A2[nc - 1];
        for (int i = nc - 2; i >= 0; i--) {
            const double* ppAij = A + i * nc + (i + 1);
            double sum = 0;
            for (int j = i + 1; j < nc; j++) { sum += *ppAij * X[j]; ppAij++; }
            X[i] = (b[i] - sum) / A2[i];
        }
    }
    static void gauss_newton(const double* L, const double* rho, double* betas) {
        for (int k = 0; k < 5; k++) {
            double A[24], b[6], x[4];
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
            qr_solve(A, b, x);
            for (int i = 0; i < 4; i++) betas[i] += x[i];
        }
    }

    void compute_pose(double Rout[9], double tout[3]) {
        choose_control_points();
        compute_barycentric_coordinates();
        // M^T M accumulated row by row (cvMulTransposed order)
        double MtM[144];
        for (int i = 0; i < 144; i++) MtM[i] = 0;
        std::vector<double> M((size_t)2 * n * 12);
        for (int i = 0; i < n; i++) {
            const double* as = &alphas[4 * i];
            double* M1 = &M[(size_t)(2 * i) * 12];
            double* M2 = M1 + 12;
            const double u = us[2 * i], v = us[2 * i + 1];
            for (int k = 0; k < 4; k++) {
                M1[3 * k] = as[k] * fu; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (uc - u);
                M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * fv; M2[3 * k + 2] = as[k] * (vc - v);
            }
        }
        for (int a = 0; a < 12; a++)
            for (int b = a; b < 12; b++) {
                double acc = 0;
                for (int r = 0; r < 2 * n; r++) acc += M[(size_t)r * 12 + a] * M[(size_t)r * 12 + b];
                MtM[a * 12 + b] = MtM[b * 12 + a] = acc;
            }
        double w[12], V[144], v4[48];
        jacobi_eig_parallel(MtM, 12, w, V);
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 12; k++) v4[i * 12 + k] = V[k * 12 + i];   // i-th smallest eigenvector
        double L[60], rho[6];
        compute_L_6x10(v4, L);
        compute_rho(rho);
        double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
        find_betas_approx_1(L, rho, Betas[1]); gauss_newton(L, rho, Betas[1]); rep[1] = compute_R_and_t(v4, Betas[1], Rs[1], ts[1]);
        find_betas_approx_2(L, rho, Betas[2]); gauss_newton(L, rho, Betas[2]); rep[2] = compute_R_and_t(v4, Betas[2], Rs[2], ts[2]);
        find_betas_approx_3(L, rho, Betas[3]); gauss_newton(L, rho, Betas[3]); rep[3] = compute_R_and_t(v4, Betas[3], Rs[3], ts[3]);
        int N = 1;
        if (rep[2] < rep[1]) N = 2;
        if (rep[3] < rep[N]) N = 3;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rout[i * 3 + j] = Rs[N][i][j]; tout[i] = ts[N][i]; }
    }
};

// PnPRansacCallback::computeError: float32 squared reprojection distance
void reproj_errors(const double rvec[3], const double tvec[3], const double K[9], const float* obj, const float* img,
                          int m, float* err) {
    double R[9];
    rodrigues_v2m(rvec, R);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    for (int i = 0; i < m; i++) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
        const float dx = img[2 * i] - px, dy = img[2 * i + 1] - py;
        err[i] = dx * dx + dy * dy;
    }
}

static int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : cv_round(num / denom);
}

// cvFindExtrinsicCameraParams2(useExtrinsicGuess=true): CvLevMarq on [rvec, tvec]
static void refine_lm(const double* obj, const double* img, int n, const double K[9], double rvec[3], double tvec[3]) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double param[6] = {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]}, prev[6];
    double JtJ[36], JtErr[6];
    auto project = [&](const double* p, bool jac, double& errnorm2) {
        if (jac) { for (int i = 0; i < 36; i++) JtJ[i] = 0; for (int i = 0; i < 6; i++) JtErr[i] = 0; }
        errnorm2 = 0;
        for (int i = 0; i < n; i++) {
            double Xc[3], dpdw[9], Rm[9];
            angle_axis_rotate(p, obj + 3 * i, Xc, dpdw, Rm);
            Xc[0] += p[3]; Xc[1] += p[4]; Xc[2] += p[5];
            const double z = Xc[2] ? 1. / Xc[2] : 1;
            const double x = Xc[0] * z, y = Xc[1] * z;
            const double ex = x * fx + cx - img[2 * i], ey = y * fy + cy - img[2 * i + 1];
            errnorm2 += ex * ex + ey * ey;
            if (jac) {
                double Ju[6], Jv[6];
                const double du[3] = {fx * z, 0, -fx * x * z}, dv[3] = {0, fy * z, -fy * y * z};
                for (int k = 0; k < 3; k++) {
                    Ju[k] = du[0] * dpdw[k] + du[1] * dpdw[3 + k] + du[2] * dpdw[6 + k];
                    Jv[k] = dv[0] * dpdw[k] + dv[1] * dpdw[3 + k] + dv[2] * dpdw[6 + k];
                    Ju[3 + k] = du[k];
                    Jv[3 + k] = dv[k];
                }
                for (int a = 0; a < 6; a++) {
                    for (int b = 0; b < 6; b++) JtJ[a * 6 + b] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
                    JtErr[a] += Ju[a] * ex + Jv[a] * ey;
                }
            }
        }
    };
    auto step = [&](int lambdaLg10) {
        const double lambda = std::exp(lambdaLg10 * std::log(10.));
        double A[36], b[6];
        memcpy(A, JtJ, sizeof(A));
        memcpy(b, JtErr, sizeof(b));
        for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
        if (!gauss_solve(A, b, 6)) for (int i = 0; i < 6; i++) b[i] = 0;
        for (int i = 0; i < 6; i++) param[i] = prev[i] - b[i];
    };
    int lambdaLg10 = -3, iters = 0;
    double prevErr2 = 0, err2 = 0;
    const int max_iter = 20;
    const double eps = FLT_EPSILON;
    // state machine of CvLevMarq::update (STARTED -> CALC_J -> CHECK_ERR ...)
    project(param, true, err2);
    memcpy(prev, param, sizeof(prev));
    step(lambdaLg10);
    prevErr2 = err2;   // iters == 0: prevErrNorm = norm(err at the initial parameters)
    for (;;) {
        project(param, false, err2);
        if (err2 > prevErr2) {   // errNorm > prevErrNorm (norms are non-negative, compare squares)
            if (++lambdaLg10 <= 16) { step(lambdaLg10); continue; }
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; i++) { dn += (param[i] - prev[i]) * (param[i] - prev[i]); pn += prev[i] * prev[i]; }
        if (++iters >= max_iter || std::sqrt(dn) / std::sqrt(pn) < eps) break;   // cvNorm(param, prevParam, CV_RELATIVE_L2)
        prevErr2 = err2;
        project(param, true, err2);
        memcpy(prev, param, sizeof(prev));
        step(lambdaLg10);
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
}

// returns number of inliers (0 on failure); rvec/tvec in/out
int pnp_ransac(const float* obj, const float* img, int m, const double K[9], double rvec[3], double tvec[3], int max_iters,
               float reproj_err, double confidence, int* inliers, int* hyp_used) {
    if (hyp_used) *hyp_used = 0;
    if (m < 5) return -1;
    const int modelPoints = 5;
    std::vector<float> err(m);
    std::vector<uint8_t> mask(m), best_mask(m, 0);
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0}, last_r[3] = {rvec[0], rvec[1], rvec[2]}, last_t[3] = {tvec[0], tvec[1], tvec[2]};
    int maxGood = 0;
    const float thr = (float)((double)reproj_err * (double)reproj_err);
    RNG rng((uint64_t)-1);
    int niters = std::max(max_iters, 1);
    auto run_kernel = [&](const int* idx, int cnt, double* r, double* t) {
        EPnP e(K, obj, img, idx, cnt);
        double R[9];
        e.compute_pose(R, t);
        rodrigues_m2v(R, r);
    };
    if (m == modelPoints) {
        run_kernel(nullptr, m, last_r, last_t);
        memcpy(best_r, last_r, sizeof(best_r)); memcpy(best_t, last_t, sizeof(best_t));
        std::fill(best_mask.begin(), best_mask.end(), 1);
        maxGood = m;
        if (hyp_used) *hyp_used = 1;
    } else {
        for (int iter = 0; iter < niters; iter++) {
            int idx[5];
            for (int i = 0; i < modelPoints;) {   // getSubset
                int idx_i;
                for (;;) {
                    idx_i = idx[i] = rng.uniform(0, m);
                    int j = 0;
                    for (; j < i; j++) if (idx_i == idx[j]) break;
                    if (j == i) break;
                }
                i++;
            }
            run_kernel(idx, modelPoints, last_r, last_t);
            if (hyp_used) (*hyp_used)++;
            reproj_errors(last_r, last_t, K, obj, img, m, err.data());
            int good = 0;
            for (int i = 0; i < m; i++) { const int f = err[i] <= thr; mask[i] = (uint8_t)f; good += f; }
            if (good > std::max(maxGood, modelPoints - 1)) {
                std::swap(mask, best_mask);
                memcpy(best_r, last_r, sizeof(best_r)); memcpy(best_t, last_t, sizeof(best_t));
                maxGood = good;
                niters = ransac_update_num_iters(confidence, (double)(m - good) / m, modelPoints, niters);
            }
        }
    }
    if (maxGood <= 0) return 0;   // rvec/tvec keep the last evaluated model, as the aliased buffers would
    std::vector<double> oi, ii;
    int n = 0;
    for (int i = 0; i < m; i++)
        if (best_mask[i]) {
            oi.push_back(obj[3 * i]); oi.push_back(obj[3 * i + 1]); oi.push_back(obj[3 * i + 2]);
            ii.push_back(img[2 * i]); ii.push_back(img[2 * i + 1]);
            inliers[n++] = i;
        }
    // refit from the last evaluated model (see FIXED CHOICES)
    memcpy(rvec, last_r, sizeof(last_r)); memcpy(tvec, last_t, sizeof(last_t));
    refine_lm(oi.data(), ii.data(), n, K, rvec, tvec);
    return n;
}

}  // namespace orc

extern "C" {
void orc_rodrigues_v2m(const double* r, double* R) { orc::rodrigues_v2m(r, R); }
void orc_rodrigues_m2v(const double* R, double* r) { orc::rodrigues_m2v(R, r); }
void orc_epnp(const float* obj, const float* img, int n, const double* K, double* R, double* t) {
    orc::EPnP e(K, obj, img, nullptr, n);
    e.compute_pose(R, t);
}
int orc_pnp_ransac(const float* obj, const float* img, int m, const double* K, double* rvec, double* tvec, int iters,
                   float reproj_err, double confidence, int* inliers, int* hyp_used) {
    return orc::pnp_ransac(obj, img, m, K, rvec, tvec, iters, reproj_err, confidence, inliers, hyp_used);
}
// debug/parity: all `iters` hypotheses without the adaptive cut-off: models (iters x 6), inlier counts
void orc_pnp_hypotheses(const float* obj, const float* img, int m, const double* K, int iters, float reproj_err, double* models, int* counts) {
    orc::RNG rng((uint64_t)-1);
    std::vector<float> err(m);
    const float thr = (float)((double)reproj_err * (double)reproj_err);
    for (int it = 0; it < iters; it++) {
        int idx[5];
        for (int i = 0; i < 5;) {
            int idx_i;
            for (;;) {
                idx_i = idx[i] = rng.uniform(0, m);
                int j = 0;
                for (; j < i; j++) if (idx_i == idx[j]) break;
                if (j == i) break;
            }
            i++;
        }
        orc::EPnP e(K, obj, img, idx, 5);
        double R[9];
        e.compute_pose(R, models + it * 6 + 3);
        orc::rodrigues_m2v(R, models + it * 6);
        orc::reproj_errors(models + it * 6, models + it * 6 + 3, K, obj, img, m, err.data());
        int good = 0;
        for (int i = 0; i < m; i++) good += err[i] <= thr;
        counts[it] = good;
    }
}
void orc_rng_sequence(uint64_t seed, int n, int bound, int* out) {
    orc::RNG r(seed);
    for (int i = 0; i < n; i++) out[i] = r.uniform(0, bound);
}
}
