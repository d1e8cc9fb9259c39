// gfx950 five-point RANSAC hypotheses for the triangulator (cv::findEssentialMat, OpenCVFivePointTri.cpp:24).
// k_fivepoint_hyp: one WORKGROUP per hypothesis, lane 0 runs EMEstimatorCallback::runKernel (Nister's solver: 5x9 null space, the ten cubic
// constraints, 10x20 elimination, det B(z) -> degree-10 polynomial, Durand-Kerner roots, back-substitution) statement for
// statement as host/vo_fivepoint.cpp:five_point_kernel does, so every model is bit-identical to the host's (IEEE +,-,*,/ and
// sqrt only, -ffp-contract=off on both sides). Every array of the solver sits in the workgroup's LDS (as locals they were scratch
// memory), the loops are not unrolled (14 KB of code instead of 71). One hypothesis is a serial chain of ~10^5 dependent FP64 operations:
// 0.66 ms for a round whose samples all converge, against ~11-20 us per sample on a host core - and a batched round ends with its SLOWEST
// sample: Durand-Kerner runs into its 300-sweep cap for the occasional polynomial (11 x the usual 27 sweeps), nearly every round of
// ~1800 samples holds one, so the rounds take 6.3 ms (measured) and the host stays the faster place for this solver in
// both uses. The kernel exists as the measurement behind that statement and as the bit-exact device form of §8f #1.
// k_fivepoint_score: Sampson error of every model on every correspondence (float32, <= thr) -> inlier counts; the sequential
// RANSAC bookkeeping (best-so-far, adaptive iteration count) is replayed by the host in sample order, as for PnP.
#include "pmv_ctx.h"
#include "backend.h"
#include "pmv_prof.h"
#include "pmv_dense.h"
#include <float.h>

namespace pmv {

namespace {
// Nister's column order: x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
__device__ const signed char FP_MONO3[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                                                {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
__device__ const signed char FP_MONO2[10][3] = {{2, 0, 0}, {0, 2, 0}, {0, 0, 2}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
__device__ const signed char FP_MONO1[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};

struct FPTables { signed char t11[4][4], t21[10][4]; };
__device__ inline void fp_tables(FPTables& T) {
    #pragma unroll 1
    for (int a = 0; a < 4; a++)
        #pragma unroll 1
        for (int b = 0; b < 4; b++) {
            const int e0 = FP_MONO1[a][0] + FP_MONO1[b][0], e1 = FP_MONO1[a][1] + FP_MONO1[b][1], e2 = FP_MONO1[a][2] + FP_MONO1[b][2];
            #pragma unroll 1
            for (int k = 0; k < 10; k++) if (FP_MONO2[k][0] == e0 && FP_MONO2[k][1] == e1 && FP_MONO2[k][2] == e2) T.t11[a][b] = (signed char)k;
        }
    #pragma unroll 1
    for (int a = 0; a < 10; a++)
        #pragma unroll 1
        for (int b = 0; b < 4; b++) {
            const int e0 = FP_MONO2[a][0] + FP_MONO1[b][0], e1 = FP_MONO2[a][1] + FP_MONO1[b][1], e2 = FP_MONO2[a][2] + FP_MONO1[b][2];
            #pragma unroll 1
            for (int k = 0; k < 20; k++) if (FP_MONO3[k][0] == e0 && FP_MONO3[k][1] == e1 && FP_MONO3[k][2] == e2) T.t21[a][b] = (signed char)k;
        }
}
__device__ inline void fp_mac11(const FPTables& T, double* r /*10*/, const double* a /*4*/, const double* b /*4*/, double s) {
    #pragma unroll 1
    for (int i = 0; i < 4; i++)
        #pragma unroll 1
        for (int j = 0; j < 4; j++) r[T.t11[i][j]] += s * (a[i] * b[j]);
}
__device__ inline void fp_mac21(const FPTables& T, double* r /*20*/, const double* a /*10*/, const double* b /*4*/, double s) {
    #pragma unroll 1
    for (int i = 0; i < 10; i++)
        #pragma unroll 1
        for (int j = 0; j < 4; j++) r[T.t21[i][j]] += s * (a[i] * b[j]);
}

// cv::solvePoly (Durand-Kerner) as restated in vo_fivepoint.cpp:solve_poly; returns the degree actually solved
__device__ inline int fp_solve_poly(const double* coeffs_in, int n0, double* rre, double* rim, double* ws /* 22 doubles */) {
    double* cre = ws; double* cim = ws + 11;
    #pragma unroll 1
    for (int i = 0; i <= n0; i++) { cre[i] = coeffs_in[i]; cim[i] = 0; }
    int n = n0;
    #pragma unroll 1
    for (; n > 1; n--) if (fabs(cre[n]) + fabs(cim[n]) > DBL_EPSILON) break;
    double pre = 1, pim = 0;
    #pragma unroll 1
    for (int i = 0; i < n; i++) {
        rre[i] = pre; rim[i] = pim;
        const double tre = pre * 1.0 - pim * 1.0, tim = pre * 1.0 + pim * 1.0;   // p = p * (1 + i)
        pre = tre; pim = tim;
    }
    double best_diff = DBL_MAX;
    int since_best = 0;
    #pragma unroll 1
    for (int iter = 0; iter < 300; iter++) {
        double maxDiff = 0;
        #pragma unroll 1
        for (int i = 0; i < n; i++) {
            pre = rre[i]; pim = rim[i];
            double nre = cre[n], nim = cim[n], dre = cre[n], dim = cim[n];
            #pragma unroll 1
            for (int j = 0; j < n; j++) {
                const double t0 = nre * pre - nim * pim, t1 = nre * pim + nim * pre;   // num = num * p
                nre = t0; nim = t1;
                nre += cre[n - j - 1]; nim += cim[n - j - 1];
                if (j != i) {
                    const double qre = pre - rre[j], qim = pim - rim[j];
                    if (qre != 0 || qim != 0) {
                        const double u0 = dre * qre - dim * qim, u1 = dre * qim + dim * qre;   // denom = denom * d
                        dre = u0; dim = u1;
                    }
                }
            }
            const double t = 1. / (dre * dre + dim * dim);
            const double qre = (nre * dre + nim * dim) * t, qim = (-nre * dim + nim * dre) * t;   // num / denom
            rre[i] = pre - qre; rim[i] = pim - qim;
            const double md = sqrt(qre * qre + qim * qim);
            maxDiff = md > maxDiff ? md : maxDiff;
        }
        double scale = 0;
        #pragma unroll 1
        for (int i = 0; i < n; i++) { const double v = fabs(rre[i]) + fabs(rim[i]); scale = v > scale ? v : scale; }
        const double lim = scale > 1.0 ? scale : 1.0;
        if (maxDiff <= 1e-14 * lim) break;
        if (maxDiff < 0.5 * best_diff) { best_diff = maxDiff; since_best = 0; }   // (the stall rule of vo_fivepoint.cpp:solve_poly)
        else if (maxDiff <= 1e-6 * lim && ++since_best >= 10) break;
    }
    #pragma unroll 1
    for (int i = 0; i < n; i++) if (fabs(rim[i]) < 1e-100) rim[i] = 0;
    return n;
}

// five normalised correspondences -> up to 10 essential matrices (row-major, unit Frobenius norm); vo_fivepoint.cpp:five_point_kernel
// Every array of the solver lives in `ws` (FP_WS doubles of LDS per hypothesis): as locals they are indexed dynamically, i.e. scratch memory -
// a round trip to L2/HBM per element access, 7 ms per hypothesis, which is what made the device form useless in round 2.
constexpr int FP_WS = 770;
__device__ int fp_essentials(const double* q1, const double* q2, double* E_out, double* ws) {
    FPTables& T = *(FPTables*)ws;
    fp_tables(T);
    double (*Q)[9] = (double (*)[9])(ws + 8);
    #pragma unroll 1
    for (int i = 0; i < 5; i++) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        Q[i][0] = x1 * x2; Q[i][1] = y1 * x2; Q[i][2] = x2; Q[i][3] = x1 * y2; Q[i][4] = y1 * y2; Q[i][5] = y2; Q[i][6] = x1; Q[i][7] = y1; Q[i][8] = 1.0;
    }
    int* colperm = (int*)(ws + 54);
    #pragma unroll 1
    for (int c = 0; c < 9; c++) colperm[c] = c;
    #pragma unroll 1
    for (int r = 0; r < 5; r++) {
        int pr = r, pc = r;
        double best = -1;
        #pragma unroll 1
        for (int i = r; i < 5; i++)
            #pragma unroll 1
            for (int j = r; j < 9; j++) if (fabs(Q[i][j]) > best) { best = fabs(Q[i][j]); pr = i; pc = j; }
        if (!(best > 1e-300)) return 0;
        if (pr != r) for (int j = 0; j < 9; j++) { const double t = Q[r][j]; Q[r][j] = Q[pr][j]; Q[pr][j] = t; }
        if (pc != r) { for (int i = 0; i < 5; i++) { const double t = Q[i][r]; Q[i][r] = Q[i][pc]; Q[i][pc] = t; } const int t = colperm[r]; colperm[r] = colperm[pc]; colperm[pc] = t; }
        const double inv = 1.0 / Q[r][r];
        #pragma unroll 1
        for (int j = 0; j < 9; j++) Q[r][j] *= inv;
        #pragma unroll 1
        for (int i = 0; i < 5; i++) {
            if (i == r) continue;
            const double f = Q[i][r];
            if (f == 0.0) continue;
            #pragma unroll 1
            for (int j = 0; j < 9; j++) Q[i][j] -= f * Q[r][j];
        }
    }
    double (*EE)[9] = (double (*)[9])(ws + 60);
    #pragma unroll 1
    for (int k = 0; k < 4; k++) {
        double* v = ws + 96;
        #pragma unroll 1
        for (int j = 0; j < 9; j++) v[j] = 0;
        #pragma unroll 1
        for (int i = 0; i < 5; i++) v[i] = -Q[i][5 + k];
        v[5 + k] = 1.0;
        double nrm = 0;
        #pragma unroll 1
        for (int j = 0; j < 9; j++) nrm += v[j] * v[j];
        nrm = sqrt(nrm);
        #pragma unroll 1
        for (int j = 0; j < 9; j++) EE[k][colperm[j]] = v[j] / nrm;
    }
    double (*Ep)[4] = (double (*)[4])(ws + 106);
    #pragma unroll 1
    for (int k = 0; k < 9; k++) { Ep[k][0] = EE[0][k]; Ep[k][1] = EE[1][k]; Ep[k][2] = EE[2][k]; Ep[k][3] = EE[3][k]; }
#define EP(r, c) Ep[(r) * 3 + (c)]
    double (*A)[20] = (double (*)[20])(ws + 142);   // the ten cubic constraints (eqs), then eliminated in place
    #pragma unroll 1
    for (int r = 0; r < 10; r++) for (int c = 0; c < 20; c++) A[r][c] = 0;
    {   // det(E) = 0
        double* m0 = ws + 342; double* m1 = ws + 352; double* m2 = ws + 362;
        #pragma unroll 1
        for (int k = 0; k < 10; k++) m0[k] = m1[k] = m2[k] = 0;
        fp_mac11(T, m0, EP(1, 1), EP(2, 2), 1.0); fp_mac11(T, m0, EP(1, 2), EP(2, 1), -1.0);
        fp_mac11(T, m1, EP(1, 0), EP(2, 2), 1.0); fp_mac11(T, m1, EP(1, 2), EP(2, 0), -1.0);
        fp_mac11(T, m2, EP(1, 0), EP(2, 1), 1.0); fp_mac11(T, m2, EP(1, 1), EP(2, 0), -1.0);
        fp_mac21(T, A[0], m0, EP(0, 0), 1.0); fp_mac21(T, A[0], m1, EP(0, 1), -1.0); fp_mac21(T, A[0], m2, EP(0, 2), 1.0);
    }
    {   // 2 E E^T E - trace(E E^T) E = 0
        double (*EEt)[10] = (double (*)[10])(ws + 372); double* tr = ws + 462;
        #pragma unroll 1
        for (int i = 0; i < 9; i++) for (int k = 0; k < 10; k++) EEt[i][k] = 0;
        #pragma unroll 1
        for (int i = 0; i < 3; i++)
            #pragma unroll 1
            for (int j = 0; j < 3; j++)
                #pragma unroll 1
                for (int k = 0; k < 3; k++) fp_mac11(T, EEt[i * 3 + j], EP(i, k), EP(j, k), 1.0);
        #pragma unroll 1
        for (int k = 0; k < 10; k++) tr[k] = EEt[0][k] + EEt[4][k] + EEt[8][k];
        #pragma unroll 1
        for (int i = 0; i < 3; i++)
            #pragma unroll 1
            for (int j = 0; j < 3; j++) {
                double* e = A[1 + i * 3 + j];
                #pragma unroll 1
                for (int k = 0; k < 3; k++) fp_mac21(T, e, EEt[i * 3 + k], EP(k, j), 2.0);
                fp_mac21(T, e, tr, EP(i, j), -1.0);
            }
    }
#undef EP
    // A <- A[:, :10]^-1 A[:, 10:]  (Gauss-Jordan with partial pivoting)
    #pragma unroll 1
    for (int c = 0; c < 10; c++) {
        int piv = c;
        #pragma unroll 1
        for (int r = c + 1; r < 10; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (fabs(A[piv][c]) < 1e-300) return 0;
        if (piv != c) for (int k = 0; k < 20; k++) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
        const double inv = 1.0 / A[c][c];
        #pragma unroll 1
        for (int k = 0; k < 20; k++) A[c][k] *= inv;
        #pragma unroll 1
        for (int r = 0; r < 10; r++) {
            if (r == c) continue;
            const double f = A[r][c];
            if (f == 0.0) continue;
            #pragma unroll 1
            for (int k = 0; k < 20; k++) A[r][k] -= f * A[c][k];
        }
    }
    double* b = ws + 472;
    #pragma unroll 1
    for (int i = 0; i < 3; i++) {
        const double* a1 = &A[i * 2 + 4][10];
        const double* a2 = &A[i * 2 + 5][10];
        double* row1 = ws + 512; double* row2 = ws + 525;
        #pragma unroll 1
        for (int k = 0; k < 13; k++) row1[k] = row2[k] = 0;
        #pragma unroll 1
        for (int k = 0; k < 3; k++) { row1[1 + k] = a1[k]; row1[5 + k] = a1[3 + k]; row2[k] = a2[k]; row2[4 + k] = a2[3 + k]; }
        #pragma unroll 1
        for (int k = 0; k < 4; k++) { row1[9 + k] = a1[6 + k]; row2[8 + k] = a2[6 + k]; }
        #pragma unroll 1
        for (int k = 0; k < 13; k++) b[i * 13 + k] = row1[k] - row2[k];
    }
    double* c = ws + 538;
    #pragma unroll 1
    for (int k = 0; k < 11; k++) c[k] = 0;
    const int perms[6][4] = {{0, 1, 2, 1}, {1, 2, 0, 1}, {2, 0, 1, 1}, {2, 1, 0, -1}, {1, 0, 2, -1}, {0, 2, 1, -1}};
    #pragma unroll 1
    for (int pi = 0; pi < 6; pi++) {
        double (*p)[5] = (double (*)[5])(ws + 550);
        int* d = (int*)(ws + 566);
        #pragma unroll 1
        for (int row = 0; row < 3; row++) {
            const double* br = b + row * 13;
            const int col = perms[pi][row];
            if (col == 0) { p[row][0] = br[3]; p[row][1] = br[2]; p[row][2] = br[1]; p[row][3] = br[0]; d[row] = 3; }
            else if (col == 1) { p[row][0] = br[7]; p[row][1] = br[6]; p[row][2] = br[5]; p[row][3] = br[4]; d[row] = 3; }
            else { p[row][0] = br[12]; p[row][1] = br[11]; p[row][2] = br[10]; p[row][3] = br[9]; p[row][4] = br[8]; d[row] = 4; }
        }
        double* t01 = ws + 568;
        #pragma unroll 1
        for (int k = 0; k < 9; k++) t01[k] = 0;
        #pragma unroll 1
        for (int i = 0; i <= d[0]; i++) for (int j = 0; j <= d[1]; j++) t01[i + j] += p[0][i] * p[1][j];
        #pragma unroll 1
        for (int i = 0; i <= d[0] + d[1]; i++) for (int j = 0; j <= d[2]; j++) c[i + j] += perms[pi][3] * t01[i] * p[2][j];
    }
    double* rre = ws + 578; double* rim = ws + 588;
    const int nroots = fp_solve_poly(c, 10, rre, rim, ws + 638);
    int count = 0;
    #pragma unroll 1
    for (int ri = 0; ri < nroots; ri++) {
        if (fabs(rim[ri]) > 1e-10) continue;
        const double z1 = rre[ri], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double* bz = ws + 598;
        #pragma unroll 1
        for (int j = 0; j < 3; j++) {
            const double* br = b + j * 13;
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double* BtB = ws + 607; double* ew = ws + 616; double* eV = ws + 619;
        #pragma unroll 1
        for (int a = 0; a < 3; a++)
            #pragma unroll 1
            for (int bb = 0; bb < 3; bb++) BtB[a * 3 + bb] = bz[a] * bz[bb] + bz[3 + a] * bz[3 + bb] + bz[6 + a] * bz[6 + bb];
        d_jacobi_eig<3>(BtB, ew, eV);
        const double xy1[3] = {eV[0], eV[3], eV[6]};
        if (fabs(xy1[2]) < 1e-10) continue;
        const double xs = xy1[0] / xy1[2], ys = xy1[1] / xy1[2];
        double* Ev = ws + 628; double nrm = 0;
        #pragma unroll 1
        for (int k = 0; k < 9; k++) { Ev[k] = EE[0][k] * xs + EE[1][k] * ys + EE[2][k] * z1 + EE[3][k]; nrm += Ev[k] * Ev[k]; }
        nrm = sqrt(nrm);
        #pragma unroll 1
        for (int k = 0; k < 9; k++) E_out[count * 9 + k] = Ev[k] / nrm;
        count++;
        if (count == 10) break;
    }
    return count;
}
}  // namespace

// samples: n_hyp x 5 correspondence indices; q1, q2: n normalised points; models: n_hyp x 90 doubles; n_models: n_hyp ints
__global__ __launch_bounds__(64) void k_fivepoint_hyp(const FivePointProblem* __restrict__ probs) { BACKEND_PRIO();
    // one workgroup = one hypothesis, solved by lane 0 on the workgroup's LDS workspace (the solver is one dependency chain; the other lanes
    // would only add divergence). 6 KB of LDS per workgroup: 26 fit a CU, a round of 62 requests x 32 hypotheses is resident at once.
    __shared__ double ws[FP_WS];
    if (threadIdx.x != 0) return;
    const FivePointProblem P = probs[blockIdx.y];
    const int h = blockIdx.x;
    if (h >= P.n_hyp) return;
    double* s1 = ws + 750; double* s2 = ws + 760;
    #pragma unroll 1
    for (int i = 0; i < 5; i++) {
        const int idx = P.samples[h * 5 + i];
        s1[2 * i] = P.q1[2 * idx]; s1[2 * i + 1] = P.q1[2 * idx + 1];
        s2[2 * i] = P.q2[2 * idx]; s2[2 * i + 1] = P.q2[2 * idx + 1];
    }
    double* E = ws + 660;
    const int nm = fp_essentials(s1, s2, E, ws);
    P.n_models_d[h] = nm; P.n_models_h[h] = nm;   // device copy for the scoring kernel, mapped pinned copy for the host
    #pragma unroll 1
    for (int k = 0; k < nm * 9; k++) { P.models_d[(size_t)h * 90 + k] = E[k]; P.models_h[(size_t)h * 90 + k] = E[k]; }
}

// EMEstimatorCallback::computeError (Sampson distance stored as float32) + inlier count per (hypothesis, model): one wavefront each
__global__ __launch_bounds__(64) void k_fivepoint_score(const FivePointProblem* __restrict__ probs) { BACKEND_PRIO();
    const FivePointProblem P = probs[blockIdx.z];
    const int h = blockIdx.y, mi = blockIdx.x;
    if (h >= P.n_hyp || mi >= P.n_models_d[h]) return;
    double E[9];
    #pragma unroll 1
    for (int k = 0; k < 9; k++) E[k] = P.models_d[(size_t)h * 90 + mi * 9 + k];
    int good = 0;
    #pragma unroll 1
    for (int i = threadIdx.x; i < P.n; i += 64) {
        const double x1[3] = {P.q1[2 * i], P.q1[2 * i + 1], 1.}, x2[3] = {P.q2[2 * i], P.q2[2 * i + 1], 1.};
        const double Ex1[3] = {E[0] * x1[0] + E[1] * x1[1] + E[2] * x1[2], E[3] * x1[0] + E[4] * x1[1] + E[5] * x1[2], E[6] * x1[0] + E[7] * x1[1] + E[8] * x1[2]};
        const double Etx2[3] = {E[0] * x2[0] + E[3] * x2[1] + E[6] * x2[2], E[1] * x2[0] + E[4] * x2[1] + E[7] * x2[2], E[2] * x2[0] + E[5] * x2[1] + E[8] * x2[2]};
        const double x2tEx1 = x2[0] * Ex1[0] + x2[1] * Ex1[1] + x2[2] * Ex1[2];
        const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = Etx2[0] * Etx2[0], d = Etx2[1] * Etx2[1];
        const float err = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
        good += err <= P.thr;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) good += __shfl_xor(good, o, 64);
    if (threadIdx.x == 0) P.counts_h[h * 10 + mi] = good;
}

hipError_t launch_fivepoint_batch(hipStream_t s, const FivePointProblem* d_probs, int n_probs, int max_hyp) {
    if (n_probs <= 0 || max_hyp <= 0) return hipSuccess;
    if (!d_probs) return hipErrorInvalidValue;
    ProfScope ps(K_FIVEPOINT, s);
    hipLaunchKernelGGL(k_fivepoint_hyp, dim3(max_hyp, n_probs), dim3(64), 0, s, d_probs);
    hipLaunchKernelGGL(k_fivepoint_score, dim3(10, max_hyp, n_probs), dim3(64), 0, s, d_probs);
    return hipGetLastError();
}

}  // namespace pmv
