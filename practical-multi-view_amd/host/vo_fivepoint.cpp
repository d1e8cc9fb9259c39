// Host triangulator (BaseTriangulator role): restates cv::findEssentialMat(p1, p2, K, RANSAC, 0.99, 1.0, mask) +
// cv::recoverPose(E, p1, p2, K, R, t, HUGE_VAL, mask, tri) as called at /root/reference/OpenCVFivePointTri.cpp:24-26
// and the adapter logic around them (:5-54).  Runs only at (re-)initialisation, so it stays on the host for now
// (SURVEY.md §8f "next #1"); the CPU-baseline pipeline and the GPU pipeline share this one implementation.
// PARITY UNPINNED (OpenCV internals; published algorithm): Nistér five-point solver inside
// RANSACPointSetRegistrator (RNG((uint64)-1), 5-point samples, <=1000 iterations, Sampson error, threshold 1px/mean focal),
// degree-10 polynomial rooted by the Durand–Kerner iteration of cv::solvePoly, four-fold (R,t) cheirality test with DLT
// triangulation.  FIXED CHOICES: the 5x9 null space by Gauss–Jordan elimination (any basis of it gives the same E set), other
// null vectors via cyclic-Jacobi eigenvectors of A^T A; the 10x20 constraint matrix is built by explicit polynomial algebra in
// Nistér's monomial order instead of OpenCV's generated coefficient table; Durand–Kerner stops at 1e-14 relative movement, or when the
// movement has stopped shrinking (round-off floor of clustered roots: solve_poly's stall rule).
#include "vo_pipeline.h"
#include "vo_math.h"
#include <cfloat>
#include <emmintrin.h>
#include <thread>

#include <chrono>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <atomic>
namespace vo {
namespace {

struct RNG {   // cv::RNG (multiply-with-carry)
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---- polynomials in (x,y,z): degree 1 (4 coefficients), degree 2 (10), degree 3 (20, Nistér's column order) ----------
// Nistér's column order: x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
const int MONO3[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                          {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
const int MONO2[10][3] = {{2, 0, 0}, {0, 2, 0}, {0, 0, 2}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
const int MONO1[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
struct PolyTables {
    int t11[4][4];    // degree-1 x degree-1 -> index into MONO2
    int t21[10][4];   // degree-2 x degree-1 -> index into MONO3
    PolyTables() {
        for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++) {
                const int e[3] = {MONO1[a][0] + MONO1[b][0], MONO1[a][1] + MONO1[b][1], MONO1[a][2] + MONO1[b][2]};
                for (int k = 0; k < 10; k++) if (MONO2[k][0] == e[0] && MONO2[k][1] == e[1] && MONO2[k][2] == e[2]) t11[a][b] = k;
            }
        for (int a = 0; a < 10; a++)
            for (int b = 0; b < 4; b++) {
                const int e[3] = {MONO2[a][0] + MONO1[b][0], MONO2[a][1] + MONO1[b][1], MONO2[a][2] + MONO1[b][2]};
                for (int k = 0; k < 20; k++) if (MONO3[k][0] == e[0] && MONO3[k][1] == e[1] && MONO3[k][2] == e[2]) t21[a][b] = k;
            }
    }
};
const PolyTables& ptab() { static const PolyTables t; return t; }
struct P1 { double c[4]; };
struct P2 { double c[10]; P2() { for (double& v : c) v = 0; } };
struct P3 { double c[20]; P3() { for (double& v : c) v = 0; } };
inline void mac11(P2& r, const P1& a, const P1& b, double s = 1.0) {   // r += s * a * b
    const PolyTables& T = ptab();
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.c[T.t11[i][j]] += s * (a.c[i] * b.c[j]);
}
inline void mac21(P3& r, const P2& a, const P1& b, double s = 1.0) {   // r += s * a * b
    const PolyTables& T = ptab();
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) r.c[T.t21[i][j]] += s * (a.c[i] * b.c[j]);
}

// cv::solvePoly (Durand–Kerner), coeffs ascending, degree n0 <= 10; returns the number of roots, written as (re, im) pairs.
// The arithmetic is the scalar formulation's, operation for operation (products, sums and their order, one IEEE division per root), so
// the roots have the same bits; what changed is how it is issued: a complex number is one 128-bit register (two multiplies, one sign
// flip and one add per complex product instead of four multiplies and two adds), nothing is allocated, and the square root of the
// convergence test is taken once per sweep (sqrt is monotonic: max over roots of sqrt(x) = sqrt(max x)). This solver is 72 % of the
// five-point kernel, which is 58 % of the host CPU time of a batched run on distinct sequences.
namespace {
typedef __m128d cx;   // (re, im)
inline cx cx_mul(cx a, cx b) {   // (a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re)
    const cx x = _mm_mul_pd(_mm_unpacklo_pd(a, a), b);                          // (a.re*b.re, a.re*b.im)
    const cx y = _mm_mul_pd(_mm_unpackhi_pd(a, a), _mm_shuffle_pd(b, b, 1));    // (a.im*b.im, a.im*b.re)
    return _mm_add_pd(x, _mm_xor_pd(y, _mm_set_pd(0.0, -0.0)));                 // x0 + (-y0) = x0 - y0 exactly; x1 + y1
}
inline double lo(cx a) { return _mm_cvtsd_f64(a); }
inline double hi(cx a) { return _mm_cvtsd_f64(_mm_unpackhi_pd(a, a)); }
}  // namespace
int solve_poly(const double* coeffs_in, int n0, double* roots_out /* 2 * n0 */) {
    int n = n0;
    for (; n > 1; n--) if (std::abs(coeffs_in[n]) + std::abs(0.0) > DBL_EPSILON) break;
    cx roots[12];
    cx p = _mm_set_pd(0, 1);
    const cx r = _mm_set_pd(1, 1);
    for (int i = 0; i < n; i++) { roots[i] = p; p = cx_mul(p, r); }
    const cx cn = _mm_set_pd(0, coeffs_in[n]);
    const int maxIters = 300;
    double best_diff = DBL_MAX;
    int since_best = 0;
    // (measured and dropped: all Horner values of a sweep up front, four roots per AVX2 instruction - they depend only on where the roots
    // stood at the sweep's start. Fewer instructions, but out of the shadow of the denominator chains: 11.5 vs 10.9 us per sample on an idle
    // core, no difference in a loaded batched run, profiles/r03_batch_exp_an.log)
    for (int iter = 0; iter < maxIters; iter++) {
        double maxDiff2 = 0;
        for (int i = 0; i < n; i++) {
            p = roots[i];
            cx num = cn, denom = cn;
            for (int j = 0; j < n; j++) {
                num = _mm_add_pd(cx_mul(num, p), _mm_set_pd(0.0, coeffs_in[n - j - 1]));   // (re + c, im + 0.0)
                if (j != i) {
                    const cx d = _mm_sub_pd(p, roots[j]);
                    if (_mm_movemask_pd(_mm_cmpneq_pd(d, _mm_setzero_pd()))) denom = cx_mul(denom, d);
                }
            }
            // num / denom: t = 1 / (b.re^2 + b.im^2); ((a.re*b.re + a.im*b.im) * t, (-a.re*b.im + a.im*b.re) * t)
            const cx bb = _mm_mul_pd(denom, denom);
            const double t = 1. / (lo(bb) + hi(bb));
            const cx ab = _mm_mul_pd(num, denom);                              // (a.re*b.re, a.im*b.im)
            const cx c = _mm_mul_pd(num, _mm_shuffle_pd(denom, denom, 1));     // (a.re*b.im, a.im*b.re)
            const cx q = _mm_mul_pd(_mm_set_pd(-lo(c) + hi(c), lo(ab) + hi(ab)), _mm_set1_pd(t));
            roots[i] = _mm_sub_pd(p, q);
            const cx qq = _mm_mul_pd(q, q);
            maxDiff2 = std::max(maxDiff2, lo(qq) + hi(qq));
        }
        // cv::solvePoly stops only at maxDiff <= 0 (or after 300 sweeps); the iteration has converged to working precision
        // long before (FIXED CHOICE: stop once no root moved by more than 1e-14 of its magnitude)
        const double maxDiff = std::sqrt(maxDiff2);
        double scale = 0;
        for (int i = 0; i < n; i++) scale = std::max(scale, std::fabs(lo(roots[i])) + std::fabs(hi(roots[i])));
        const double lim = scale > 1.0 ? scale : 1.0;
        if (maxDiff <= 1e-14 * lim) break;
        // ... or once the movement has stopped shrinking: 2 % of the samples (clustered roots) stall at a round-off floor of 1e-8 .. 1e-13 of
        // the root magnitude from sweep 20-40 on and used to run all 300 sweeps for nothing - a fifth of all sweeps (FIXED CHOICE, round 3:
        // in the convergence regime, ten sweeps in a row without halving the smallest movement seen so far end the iteration)
        if (maxDiff < 0.5 * best_diff) { best_diff = maxDiff; since_best = 0; }
        else if (maxDiff <= 1e-6 * lim && ++since_best >= 10) break;
    }
    for (int i = 0; i < n; i++) {
        double im = hi(roots[i]);
        if (std::fabs(im) < 1e-100) im = 0;
        roots_out[2 * i] = lo(roots[i]); roots_out[2 * i + 1] = im;
    }
    return n;
}

// EMEstimatorCallback::runKernel: five normalised correspondences -> up to 10 essential matrices (row-major 3x3 each)
int five_point_kernel(const double* q1, const double* q2, double* E_out) {
    double Q[5][9];
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        Q[i][0] = x1 * x2; Q[i][1] = y1 * x2; Q[i][2] = x2; Q[i][3] = x1 * y2; Q[i][4] = y1 * y2; Q[i][5] = y2; Q[i][6] = x1; Q[i][7] = y1; Q[i][8] = 1.0;
    }
    // Null space of the 5x9 system. OpenCV takes the last four right singular vectors; the solution set {E} only depends
    // on the 4-dimensional null SPACE, not on its basis, so (FIXED CHOICE) the basis comes from Gauss–Jordan elimination with
    // full pivoting: x_free = e_k, x_pivot = -B[:, k].
    int colperm[9];
    for (int c = 0; c < 9; c++) colperm[c] = c;
    for (int r = 0; r < 5; r++) {
        int pr = r, pc = r;
        double best = -1;
        for (int i = r; i < 5; i++)
            for (int j = r; j < 9; j++) if (std::fabs(Q[i][j]) > best) { best = std::fabs(Q[i][j]); pr = i; pc = j; }
        if (!(best > 1e-300)) return 0;   // rank-deficient sample
        if (pr != r) for (int j = 0; j < 9; j++) std::swap(Q[r][j], Q[pr][j]);
        if (pc != r) { for (int i = 0; i < 5; i++) std::swap(Q[i][r], Q[i][pc]); std::swap(colperm[r], colperm[pc]); }
        const double inv = 1.0 / Q[r][r];
        for (int j = 0; j < 9; j++) Q[r][j] *= inv;
        for (int i = 0; i < 5; i++) {
            if (i == r) continue;
            const double f = Q[i][r];
            if (f == 0.0) continue;
            for (int j = 0; j < 9; j++) Q[i][j] -= f * Q[r][j];
        }
    }
    double EE[4][9];
    for (int k = 0; k < 4; k++) {
        double v[9];
        for (int j = 0; j < 9; j++) v[j] = 0;
        for (int i = 0; i < 5; i++) v[i] = -Q[i][5 + k];
        v[5 + k] = 1.0;
        double nrm = 0;
        for (int j = 0; j < 9; j++) nrm += v[j] * v[j];
        nrm = std::sqrt(nrm);
        for (int j = 0; j < 9; j++) EE[k][colperm[j]] = v[j] / nrm;
    }
    // E(x,y,z) = x EE0 + y EE1 + z EE2 + EE3 as 9 linear polynomials
    P1 Ep[9];
    for (int k = 0; k < 9; k++) { Ep[k].c[0] = EE[0][k]; Ep[k].c[1] = EE[1][k]; Ep[k].c[2] = EE[2][k]; Ep[k].c[3] = EE[3][k]; }
    auto P = [&](int r, int c) -> const P1& { return Ep[r * 3 + c]; };
    P3 eqs[10];
    // det(E) = 0
    {
        P2 m0, m1, m2;
        mac11(m0, P(1, 1), P(2, 2)); mac11(m0, P(1, 2), P(2, 1), -1.0);
        mac11(m1, P(1, 0), P(2, 2)); mac11(m1, P(1, 2), P(2, 0), -1.0);
        mac11(m2, P(1, 0), P(2, 1)); mac11(m2, P(1, 1), P(2, 0), -1.0);
        mac21(eqs[0], m0, P(0, 0)); mac21(eqs[0], m1, P(0, 1), -1.0); mac21(eqs[0], m2, P(0, 2));
    }
    // 2 E E^T E - trace(E E^T) E = 0
    {
        P2 EEt[9], tr;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) mac11(EEt[i * 3 + j], P(i, k), P(j, k));
        for (int k = 0; k < 10; k++) tr.c[k] = EEt[0].c[k] + EEt[4].c[k] + EEt[8].c[k];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                P3& e = eqs[1 + i * 3 + j];
                for (int k = 0; k < 3; k++) mac21(e, EEt[i * 3 + k], P(k, j), 2.0);
                mac21(e, tr, P(i, j), -1.0);
            }
    }
    double A[10][20];
    for (int r = 0; r < 10; r++)
        for (int c = 0; c < 20; c++) A[r][c] = eqs[r].c[c];
    // A <- A[:, :10]^-1 A[:, 10:]  (Gauss–Jordan with partial pivoting)
    for (int c = 0; c < 10; c++) {
        int piv = c;
        for (int r = c + 1; r < 10; r++) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (std::fabs(A[piv][c]) < 1e-300) return 0;
        if (piv != c) for (int k = 0; k < 20; k++) std::swap(A[c][k], A[piv][k]);
        const double inv = 1.0 / A[c][c];
        for (int k = 0; k < 20; k++) A[c][k] *= inv;
        for (int r = 0; r < 10; r++) {
            if (r == c) continue;
            const double f = A[r][c];
            if (f == 0.0) continue;
            for (int k = 0; k < 20; k++) A[r][k] -= f * A[c][k];
        }
    }
    // B(z) [x y 1]^T = 0 with rows (x^2z-row) - z (x^2-row), (y^2z) - z (y^2), (xyz) - z (xy)
    double b[3 * 13];
    for (int i = 0; i < 3; i++) {
        const double* a1 = &A[i * 2 + 4][10];
        const double* a2 = &A[i * 2 + 5][10];
        double row1[13] = {0}, row2[13] = {0};
        for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
        for (int k = 0; k < 4; k++) { row1[9 + k] = a1[6 + k]; row2[8 + k] = a2[6 + k]; }
        for (int k = 0; k < 13; k++) b[i * 13 + k] = row1[k] - row2[k];
    }
    // det B(z): polynomial of degree 10 (coefficients ascending)
    auto polyz = [&](int row, int col, double* out) -> int {   // ascending coefficients; returns degree
        const double* br = b + row * 13;
        if (col == 0) { out[0] = br[3]; out[1] = br[2]; out[2] = br[1]; out[3] = br[0]; return 3; }
        if (col == 1) { out[0] = br[7]; out[1] = br[6]; out[2] = br[5]; out[3] = br[4]; return 3; }
        out[0] = br[12]; out[1] = br[11]; out[2] = br[10]; out[3] = br[9]; out[4] = br[8];
        return 4;
    };
    double c[11];
    for (double& v : c) v = 0;
    const int perms[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}, {1, 0, 2, -1}, {0, 2, 1, -1}};
    for (auto& pm : perms) {
        double p0[5], p1[5], p2[5];
        const int d0 = polyz(0, pm[0], p0), d1 = polyz(1, pm[1], p1), d2 = polyz(2, pm[2], p2);
        double t01[9] = {0};
        for (int i = 0; i <= d0; i++) for (int j = 0; j <= d1; j++) t01[i + j] += p0[i] * p1[j];
        for (int i = 0; i <= d0 + d1; i++) for (int j = 0; j <= d2; j++) c[i + j] += pm[3] * t01[i] * p2[j];
    }
    double roots[20];
    const int n_roots = solve_poly(c, 10, roots);
    int count = 0;
    for (int ri = 0; ri < n_roots; ri++) {
        if (std::fabs(roots[2 * ri + 1]) > 1e-10) continue;
        const double z1 = roots[2 * ri], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; j++) {
            const double* br = b + j * 13;
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        // SVD::solveZ: unit null vector = eigenvector of the smallest eigenvalue of Bz^T Bz
        double BtB[9], ew[3], eV[9];
        for (int a = 0; a < 3; a++)
            for (int bb = 0; bb < 3; bb++) BtB[a * 3 + bb] = bz[a] * bz[bb] + bz[3 + a] * bz[3 + bb] + bz[6 + a] * bz[6 + bb];
        vmath::jacobi_eig(BtB, 3, ew, eV);
        const double xy1[3] = {eV[0], eV[3], eV[6]};
        if (std::fabs(xy1[2]) < 1e-10) continue;
        const double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; k++) { Ev[k] = EE[0][k] * xs + EE[1][k] * ys + EE[2][k] * z1 + EE[3][k]; nrm += Ev[k] * Ev[k]; }
        nrm = std::sqrt(nrm);
        for (int k = 0; k < 9; k++) E_out[count * 9 + k] = Ev[k] / nrm;
        count++;
        if (count == 10) break;
    }
    return count;
}

// EMEstimatorCallback::computeError (Sampson distance, stored as float32)
void sampson_errors(const double* E, const double* q1, const double* q2, int n, float* err) {
    for (int i = 0; i < n; i++) {
        const double x1[3] = {q1[2 * i], q1[2 * i + 1], 1.}, x2[3] = {q2[2 * i], q2[2 * i + 1], 1.};
        const double Ex1[3] = {E[0] * x1[0] + E[1] * x1[1] + E[2] * x1[2], E[3] * x1[0] + E[4] * x1[1] + E[5] * x1[2], E[6] * x1[0] + E[7] * x1[1] + E[8] * x1[2]};
        const double Etx2[3] = {E[0] * x2[0] + E[3] * x2[1] + E[6] * x2[2], E[1] * x2[0] + E[4] * x2[1] + E[7] * x2[2], E[2] * x2[0] + E[5] * x2[1] + E[8] * x2[2]};
        const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
        const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
        err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
    }
}

// RANSACPointSetRegistrator::getSubset: modelPoints distinct indices in [0, n), rejection sampling on the MWC stream
void draw_subset(RNG& rng, int n, int modelPoints, int* idx) {
    for (int i = 0; i < modelPoints;) {
        int idx_i;
        for (;;) {
            idx_i = idx[i] = rng.uniform(0, n);
            int j = 0;
            for (; j < i; j++) if (idx_i == idx[j]) break;
            if (j == i) break;
        }
        i++;
    }
}

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

}  // namespace

int five_point_essentials(const double* q1, const double* q2, double* E_out) { return five_point_kernel(q1, q2, E_out); }
// the first `count` 5-subsets findEssentialMat's RANSAC draws for n correspondences (cv::RNG seeded (uint64)-1 per call)
void five_point_sample_stream(int n, int count, int* out5) {
    RNG rng((uint64_t)-1);
    for (int k = 0; k < count; k++) draw_subset(rng, n, 5, out5 + 5 * k);
}
int five_point_update_num_iters(double p, double ep, int model_points, int max_iters) { return ransac_update_num_iters(p, ep, model_points, max_iters); }

// Helper threads for the five-point RANSAC: the hypotheses of a batch are independent, so they are evaluated side by side and
// the sequential bookkeeping (best-so-far, adaptive iteration count) is replayed in sample order afterwards — the outcome is
// the sequential algorithm's, whatever the thread count. Workers sleep between calls and spin only while a call is active.
class SpinPool {
public:
    explicit SpinPool(int workers) {
        for (int i = 0; i < workers; i++) th_.emplace_back([this] { worker(); });
    }
    ~SpinPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true); active_.store(true); }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void begin() { { std::lock_guard<std::mutex> lk(mu_); active_.store(true, std::memory_order_release); } cv_.notify_all(); }
    void end() { active_.store(false, std::memory_order_release); }
    // runs fn(0..n-1) on the workers and the calling thread; returns when all are done
    void run(int n, const std::function<void(int)>& fn) {
        fn_ = &fn; ntask_ = n;
        done_.store(0, std::memory_order_relaxed);
        next_.store(0, std::memory_order_release);   // a worker still leaving the previous batch may take a ticket of this one
        epoch_.fetch_add(1, std::memory_order_release);
        drain(epoch_.load(std::memory_order_relaxed));
        while (done_.load(std::memory_order_acquire) < n) { /* spin: the tasks are microseconds long */ }
    }
private:
    void drain(unsigned e) {
        for (;;) {
            if (epoch_.load(std::memory_order_acquire) != e) return;   // a late worker must not touch the next batch's counters
            const int t = next_.fetch_add(1, std::memory_order_acq_rel);
            if (t >= ntask_) break;
            (*fn_)(t);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void worker() {
        unsigned seen = epoch_.load();
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [this] { return active_.load(); }); }
            if (stop_.load()) return;
            while (active_.load(std::memory_order_acquire)) {
                if (stop_.load(std::memory_order_relaxed)) return;
                const unsigned e = epoch_.load(std::memory_order_acquire);
                if (e != seen) { seen = e; drain(e); }
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<bool> active_{false}, stop_{false};
    std::atomic<unsigned> epoch_{0};
    std::atomic<int> next_{0}, done_{0};
    int ntask_ = 0;
    const std::function<void(int)>* fn_ = nullptr;
};

// cv::findEssentialMat(points1, points2, K, RANSAC, prob, threshold, mask): returns false when no model was found
std::shared_ptr<SpinPool> make_spin_pool(int workers) { return std::make_shared<SpinPool>(workers); }

bool find_essential_mat(const double* p1, const double* p2, int n, const double* K, double prob, double threshold,
                        double* E, std::vector<uint8_t>& mask, int* samples_drawn, SpinPool* pool, int pool_width, FivePointTri* hook) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> q1(2 * n), q2(2 * n);
    for (int i = 0; i < n; i++) {
        q1[2 * i] = (p1[2 * i] - cx) / fx; q1[2 * i + 1] = (p1[2 * i + 1] - cy) / fy;
        q2[2 * i] = (p2[2 * i] - cx) / fx; q2[2 * i + 1] = (p2[2 * i + 1] - cy) / fy;
    }
    threshold /= (fx + fy) / 2;
    const int modelPoints = 5;
    mask.assign(n, 0);
    if (n < modelPoints) return false;
    std::vector<float> err(n);
    std::vector<uint8_t> cur(n);
    double models[90], best[9];
    int maxGood = 0, niters = 1000;
    const float thr = (float)(threshold * threshold);
    RNG rng((uint64_t)-1);
    auto evaluate = [&](int nmodels) {
        for (int mi = 0; mi < nmodels; mi++) {
            sampson_errors(models + 9 * mi, q1.data(), q2.data(), n, err.data());
            int good = 0;
            for (int i = 0; i < n; i++) { const int f = err[i] <= thr; cur[i] = (uint8_t)f; good += f; }
            if (good > std::max(maxGood, modelPoints - 1)) {
                std::swap(cur, mask);
                memcpy(best, models + 9 * mi, sizeof(best));
                maxGood = good;
                niters = ransac_update_num_iters(prob, (double)(n - good) / n, modelPoints, niters);
            }
        }
    };
    if (n == modelPoints) {
        const int nm = five_point_kernel(q1.data(), q2.data(), models);
        if (nm <= 0) return false;
        memcpy(E, models, 9 * sizeof(double));
        mask.assign(n, 1);
        return true;
    }
    auto draw = [&](int* idx) { draw_subset(rng, n, modelPoints, idx); };
    bool hooked = false;
    if (hook) {
        // rounds of hypotheses evaluated by the plugin's kernel hook (models + inlier counts), then the sequential bookkeeping in
        // sample order — the same replay as the helper-thread path below, so the outcome is the sequential algorithm's
        const int B = FivePointTri::HYP_ROUND;
        std::vector<int> idx(5 * B), nm(B), counts(10 * B);
        std::vector<double> models_b((size_t)90 * B);
        int iter = 0;
        hooked = true;
        while (iter < niters) {
            const int nb = std::min(B, niters - iter);
            for (int b = 0; b < nb; b++) draw(&idx[5 * b]);
            if (!hook->essential_hypotheses(q1.data(), q2.data(), n, idx.data(), nb, thr, models_b.data(), nm.data(), counts.data())) {
                if (iter == 0) { hooked = false; rng = RNG((uint64_t)-1); break; }   // no kernel behind the hook: host path from a fresh stream
                return false;
            }
            for (int b = 0; b < nb && iter < niters; b++, iter++) {
                if (samples_drawn) ++*samples_drawn;
                for (int mi = 0; mi < nm[b]; mi++) {
                    if (counts[10 * b + mi] > std::max(maxGood, modelPoints - 1)) {
                        sampson_errors(&models_b[(size_t)90 * b + 9 * mi], q1.data(), q2.data(), n, err.data());
                        for (int i = 0; i < n; i++) mask[i] = (uint8_t)(err[i] <= thr);
                        memcpy(best, &models_b[(size_t)90 * b + 9 * mi], sizeof(best));
                        maxGood = counts[10 * b + mi];
                        niters = ransac_update_num_iters(prob, (double)(n - maxGood) / n, modelPoints, niters);
                    }
                }
            }
        }
    }
    if (hooked) {
    } else if (!pool || pool_width <= 1) {
        for (int iter = 0; iter < niters; iter++) {
            if (samples_drawn) ++*samples_drawn;
            int idx[5];
            draw(idx);
            double s1[10], s2[10];
            for (int i = 0; i < 5; i++) { s1[2 * i] = q1[2 * idx[i]]; s1[2 * i + 1] = q1[2 * idx[i] + 1]; s2[2 * i] = q2[2 * idx[i]]; s2[2 * i + 1] = q2[2 * idx[i] + 1]; }
            const int nm = five_point_kernel(s1, s2, models);
            if (nm <= 0) continue;
            evaluate(nm);
        }
    } else {
        // batches of independent hypotheses on the helper threads, then the sequential bookkeeping in sample order
        const int B = std::min(64, 3 * pool_width);
        struct Hyp { int idx[5]; int nm; double models[90]; int good[10]; };
        std::vector<Hyp> hyp(B);
        std::vector<std::vector<float>> errs(B, std::vector<float>(n));
        pool->begin();
        int iter = 0;
        while (iter < niters) {
            const int nb = std::min(B, niters - iter);
            for (int b = 0; b < nb; b++) draw(hyp[b].idx);   // the index stream is sequential
            const std::function<void(int)> task = [&](int b) {
                Hyp& h = hyp[b];
                double s1[10], s2[10];
                for (int i = 0; i < 5; i++) { s1[2 * i] = q1[2 * h.idx[i]]; s1[2 * i + 1] = q1[2 * h.idx[i] + 1]; s2[2 * i] = q2[2 * h.idx[i]]; s2[2 * i + 1] = q2[2 * h.idx[i] + 1]; }
                h.nm = five_point_kernel(s1, s2, h.models);
                for (int mi = 0; mi < h.nm; mi++) {
                    sampson_errors(h.models + 9 * mi, q1.data(), q2.data(), n, errs[b].data());
                    int good = 0;
                    for (int i = 0; i < n; i++) good += errs[b][i] <= thr;
                    h.good[mi] = good;
                }
            };
            pool->run(nb, task);
            for (int b = 0; b < nb && iter < niters; b++, iter++) {
                if (samples_drawn) ++*samples_drawn;
                const Hyp& h = hyp[b];
                for (int mi = 0; mi < h.nm; mi++) {
                    if (h.good[mi] > std::max(maxGood, modelPoints - 1)) {
                        sampson_errors(h.models + 9 * mi, q1.data(), q2.data(), n, err.data());
                        for (int i = 0; i < n; i++) mask[i] = (uint8_t)(err[i] <= thr);
                        memcpy(best, h.models + 9 * mi, sizeof(best));
                        maxGood = h.good[mi];
                        niters = ransac_update_num_iters(prob, (double)(n - maxGood) / n, modelPoints, niters);
                    }
                }
            }
        }
        pool->end();
    }
    if (maxGood <= 0) { mask.assign(n, 0); return false; }
    memcpy(E, best, sizeof(best));
    return true;
}

// cv::recoverPose(E, points1, points2, K, R, t, distanceThresh = HUGE_VAL, mask (in/out), triangulatedPoints 4xN)
int recover_pose(FivePointTri* self, const double* E, const double* p1, const double* p2, int n, const double* K, double* R_out,
                 double* t_out, std::vector<uint8_t>& mask, std::vector<double>& tri4, bool ahead) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> q1(2 * n), q2(2 * n);
    for (int i = 0; i < n; i++) {
        q1[2 * i] = (p1[2 * i] - cx) / fx; q1[2 * i + 1] = (p1[2 * i + 1] - cy) / fy;
        q2[2 * i] = (p2[2 * i] - cx) / fx; q2[2 * i + 1] = (p2[2 * i + 1] - cy) / fy;
    }
    // decomposeEssentialMat
    double U[9], s[3], V[9], Vt[9];
    vmath::svd3(E, U, s, V);
    vmath::mat3_T(V, Vt);
    if (vmath::det3(U) < 0) for (double& v : U) v = -v;
    if (vmath::det3(Vt) < 0) for (double& v : Vt) v = -v;
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double R1[9], R2[9], tmp[9];
    vmath::mat3_mul(U, W, tmp); vmath::mat3_mul(tmp, Vt, R1);
    vmath::mat3_mul(U, Wt, tmp); vmath::mat3_mul(tmp, Vt, R2);
    const double tv[3] = {U[2], U[5], U[8]};
    const double* Rs[4] = {R1, R2, R1, R2};
    const double tsgn[4] = {1, 1, -1, -1};
    double P1x4[48];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) P1x4[c * 12 + i * 4 + j] = Rs[c][i * 3 + j]; P1x4[c * 12 + i * 4 + 3] = tsgn[c] * tv[i]; }
    std::vector<double> Q((size_t)16 * n);
    std::vector<uint8_t> masks((size_t)4 * n);
    int good[4] = {0, 0, 0, 0};
    // the four (R, t) candidates (cv::recoverPose evaluates them one after another) go through the kernel hook
    if (ahead) self->dlt_candidates_ahead(q1.data(), q2.data(), n, P1x4, mask.data(), Q.data(), masks.data(), good);
    else self->dlt_candidates(q1.data(), q2.data(), n, P1x4, mask.data(), Q.data(), masks.data(), good);
    int sel;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) sel = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) sel = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) sel = 2;
    else sel = 3;
    memcpy(R_out, Rs[sel], 9 * sizeof(double));
    for (int i = 0; i < 3; i++) t_out[i] = tsgn[sel] * tv[i];
    mask.assign(masks.begin() + (size_t)sel * n, masks.begin() + (size_t)(sel + 1) * n);
    tri4.assign(Q.begin() + (size_t)sel * 4 * n, Q.begin() + (size_t)(sel + 1) * 4 * n);
    return good[sel];
}

void dlt_candidates_host(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                         uint8_t* out_mask, int* out_good) {
    auto eval_combo = [&](int c) {
        const double* P1 = P1x4 + 12 * c;
        const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        int good = 0;
        for (int i = 0; i < n; i++) {
            // cvTriangulatePoints: 4x4 DLT, solution = right singular vector of the smallest singular value
            double A[16];
            const double* Ps[2] = {P0, P1};
            const double xs[2] = {q1[2 * i], q2[2 * i]}, ys[2] = {q1[2 * i + 1], q2[2 * i + 1]};
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 4; k++) {
                    A[(j * 2) * 4 + k] = xs[j] * Ps[j][8 + k] - Ps[j][k];
                    A[(j * 2 + 1) * 4 + k] = ys[j] * Ps[j][8 + k] - Ps[j][4 + k];
                }
            double AtA[16], w4[4], V4[16];
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) {
                    double acc = 0;
                    for (int k = 0; k < 4; k++) acc += A[k * 4 + a] * A[k * 4 + b];
                    AtA[a * 4 + b] = acc;
                }
            vmath::jacobi_eig(AtA, 4, w4, V4);
            const double Qh[4] = {V4[0], V4[4], V4[8], V4[12]};
            for (int k = 0; k < 4; k++) out_Q[((size_t)c * 4 + k) * n + i] = Qh[k];
            bool m = Qh[2] * Qh[3] > 0;
            const double Qn[4] = {Qh[0] / Qh[3], Qh[1] / Qh[3], Qh[2] / Qh[3], Qh[3] / Qh[3]};
            m = m && (Qn[2] < HUGE_VAL);
            const double z2 = P1[8] * Qn[0] + P1[9] * Qn[1] + P1[10] * Qn[2] + P1[11] * Qn[3];
            m = m && (z2 > 0) && (z2 < HUGE_VAL);
            m = m && mask_in[i];
            out_mask[(size_t)c * n + i] = m ? 1 : 0;
            good += m ? 1 : 0;
        }
        out_good[c] = good;
    };
    if (n >= 64) {   // the candidates are independent: one host thread each
        std::thread th[3];
        for (int c = 1; c < 4; c++) th[c - 1] = std::thread(eval_combo, c);
        eval_combo(0);
        for (auto& t : th) t.join();
    } else {
        for (int c = 0; c < 4; c++) eval_combo(c);
    }
}

// ---- findEssentialMat ahead of time (see vo_pipeline.h) ----------------------------------------------------------------
void FivePointTri::prefetch(const Frame& prev, const Frame& next) {
    if (prefetch_threads <= 0 || use_hypothesis_hook) return;
    auto job = std::make_shared<EssentialJob>();
    job->frame = prev.frame;
    const FeatureCorr& fc = prev.feat_corr;
    job->p1.reserve(2 * fc.size()); job->p2.reserve(2 * fc.size());
    fc.order.for_each([&](int c) {   // the loop of triangulate() (OpenCVFivePointTri.cpp:9-22), coordinates only
        const int fst = fc.key[(size_t)c], sec = fc.val[(size_t)c];
        if (sec < 0) return;
        job->p1.push_back(prev.column[(size_t)fst]); job->p1.push_back(prev.row[(size_t)fst]);
        job->p2.push_back(next.column[(size_t)sec]); job->p2.push_back(next.row[(size_t)sec]);
    });
    std::lock_guard<std::mutex> lk(pf_mu);
    if (pf_threads.empty())
        for (int i = 0; i < prefetch_threads; i++) pf_threads.emplace_back([this] { prefetch_worker(); });
    pf_jobs[job->frame] = job;
    pf_queue.push_back(std::move(job));
    pf_cv.notify_one();
}

void FivePointTri::prefetch_worker() {
    for (;;) {
        std::shared_ptr<EssentialJob> job;
        {
            std::unique_lock<std::mutex> lk(pf_mu);
            pf_cv.wait(lk, [&] { return pf_stop || !pf_queue.empty(); });
            if (pf_stop) return;
            job = std::move(pf_queue.front());
            pf_queue.pop_front();
        }
        int expected = 0;
        if (!job->state.compare_exchange_strong(expected, 1)) continue;   // the back-end got there first
        try {
            const int n = (int)(job->p1.size() / 2);
            job->ok = find_essential_mat(job->p1.data(), job->p2.data(), n, tracker->camera, 0.99, 1.0, job->E, job->mask, &job->drawn, nullptr, 1, nullptr);
            if (job->ok) recover_pose(this, job->E, job->p1.data(), job->p2.data(), n, tracker->camera, job->R, job->t, job->mask, job->tri, true);
        } catch (...) { job->error = std::current_exception(); }
        job->state.store(2, std::memory_order_release);
    }
}

void FivePointTri::finish() {
    { std::lock_guard<std::mutex> lk(pf_mu); pf_stop = true; pf_queue.clear(); pf_jobs.clear(); }
    pf_cv.notify_all();
    for (auto& t : pf_threads) t.join();
    pf_threads.clear();
    { std::lock_guard<std::mutex> lk(pf_mu); pf_stop = false; }   // (a later run on the same object starts its helpers again)
}

FivePointTri::~FivePointTri() { finish(); }

// ---- OpenCVFivePointTri.cpp:5-54 ---------------------------------------------------------------------------------
void FivePointTri::triangulate(Frame& src, Frame& next, Mat3& R_out, Vec3& t_out) {
    const int j = src.frame;
    std::vector<double> p1, p2;
    std::vector<int> p1_ptr, p2_ptr;   // the two features of each correspondence: rows of src / next
    // a result (or a job) from the front-end's prefetch for this frame pair; older entries belong to frames that went through PnP
    std::shared_ptr<EssentialJob> job;
    if (prefetch_threads > 0) {
        std::lock_guard<std::mutex> lk(pf_mu);
        auto it = pf_jobs.find(j);
        if (it != pf_jobs.end()) job = it->second;
        for (auto k = pf_jobs.begin(); k != pf_jobs.end();) k = k->first <= j ? pf_jobs.erase(k) : std::next(k);
    }
    bool claimed = false;   // the job exists but no helper has started it: compute here, as without prefetch
    if (job) { int expected = 0; claimed = job->state.compare_exchange_strong(expected, 1); }
    const bool inline_e = !job || claimed;
    if (inline_e && workers > 1) {   // wake the helper threads now: they are spinning by the time the gather below is done
        if (!pool) pool = std::make_shared<SpinPool>(workers - 1);
        pool->begin();
    }
    HostProfScope* hpg = new HostProfScope(tracker->stats.hp.t[4]);
    {
        const FeatureCorr& fc = src.feat_corr;
        p1.reserve(2 * fc.size()); p2.reserve(2 * fc.size()); p1_ptr.reserve(fc.size()); p2_ptr.reserve(fc.size());
        fc.order.for_each([&](int c) {   // for (auto& p : src.feat_corr)
            const int fst = fc.key[(size_t)c], sec = fc.val[(size_t)c];
            if (sec < 0) return;                                                     // p.second.expired(): the empty entries of quirk Q10
            p1.push_back(src.column[(size_t)fst]); p1.push_back(src.row[(size_t)fst]);     // integer cv::Point (quirk Q12)
            p2.push_back(next.column[(size_t)sec]); p2.push_back(next.row[(size_t)sec]);
            p1_ptr.push_back(fst);
            p2_ptr.push_back(sec);
        });
    }
    delete hpg;
    const int n = (int)(p1.size() / 2);
    std::vector<uint8_t> mask;
    std::vector<double> tri;
    double E[9];
    int drawn = 0;
    auto tE = std::chrono::steady_clock::now();
    HostCpuScope* cpu_e = new HostCpuScope(tracker->stats.hp.t[14]);
    bool ok, have_pose = false;
    if (!inline_e && job->p1 == p1 && job->p2 == p2) {   // same correspondences in the same order (always, by construction)
        while (job->state.load(std::memory_order_acquire) != 2) std::this_thread::yield();
        if (job->error) { delete cpu_e; std::rethrow_exception(job->error); }
        ok = job->ok; drawn = job->drawn;
        for (int i = 0; i < 9; i++) E[i] = job->E[i];
        if (ok) {   // recoverPose ran in the helper as well: R, unit t, the mask it updated, the homogeneous points
            mask.swap(job->mask); tri.swap(job->tri);
            for (int i = 0; i < 9; i++) R_out.m[i] = job->R[i];
            for (int i = 0; i < 3; i++) t_out.v[i] = job->t[i];
            have_pose = true;
        }
        prefetch_hits++;
        tracker->stats.tri_ahead++;
    } else {
        ok = find_essential_mat(p1.data(), p2.data(), n, tracker->camera, 0.99, 1.0, E, mask, &drawn, inline_e ? pool.get() : nullptr, inline_e ? workers : 1,
                                use_hypothesis_hook ? this : nullptr);
        prefetch_inline++;
    }
    delete cpu_e;
    if (inline_e && pool) pool->end();
    tracker->stats.t_tri_essential += std::chrono::duration<double>(std::chrono::steady_clock::now() - tE).count();
    tracker->stats.tri_hypotheses += drawn;
    if (!ok) {
        // cv::recoverPose on an empty E raises in the reference (uncaught). Keep the pipeline alive: no motion, no landmarks.
        R_out = Mat3::eye();
        t_out = Vec3{{0, 0, 0}};
        return;
    }
    auto tP = std::chrono::steady_clock::now();
    if (!have_pose) recover_pose(this, E, p1.data(), p2.data(), n, tracker->camera, R_out.m, t_out.v, mask, tri);
    tracker->stats.t_tri_pose += std::chrono::duration<double>(std::chrono::steady_clock::now() - tP).count();
    const Vec3& g1 = tracker->gt_t[j + tracker->init_offset + 1];
    const Vec3& g0 = tracker->gt_t[j + tracker->init_offset];
    const double d0 = g1.v[0] - g0.v[0], d1 = g1.v[1] - g0.v[1], d2 = g1.v[2] - g0.v[2];
    tracker->scale = std::sqrt(std::pow(d0, 2) + std::pow(d1, 2) + std::pow(d2, 2));
    t_out = tracker->scale * t_out;
    HostProfScope hpl(tracker->stats.hp.t[5]);
    for (int i = 0; i < n; i++) {
        if (!mask[i]) continue;   // Removing RANSAC outliers
        const double w = tri[(size_t)3 * n + i];
        Feature3D f3d(tracker->scale * tri[i] / w, tracker->scale * tri[(size_t)n + i] / w, tracker->scale * tri[(size_t)2 * n + i] / w * -1);
        if (f3d.z < 0) {
            f3d.transform(tracker->R[j], tracker->t[j]);
            const int id = tracker->landmarks.create(f3d);   // tracker->feats3d.push_back(f3d_ptr)
            next.lm[(size_t)p2_ptr[(size_t)i]] = id;         // next.map[p2_ptr[i]] = weak_ptr(f3d_ptr)
            src.lm[(size_t)p1_ptr[(size_t)i]] = id;          // src.map[p1_ptr[i]] = weak_ptr(f3d_ptr)
        }
    }
}

}  // namespace vo
