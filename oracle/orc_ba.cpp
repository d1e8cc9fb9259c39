// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h).
//
// (1) ba_residual_jacobian: PINNED by in-repo source /root/reference/include/ProjectionResidual.h:38-58
//     (residual) — the Jacobians are the exact analytic derivative of that expression, i.e. what
//     ceres::AutoDiffCostFunction<ProjectionResidual,2,6,3> (/root/reference/ProjectionResidual.cpp:6) evaluates,
//     including ceres::AngleAxisRotatePoint's two branches (theta^2 > eps: Rodrigues; else first order).
// (2) ba_solve: PARITY UNPINNED. Restates ceres::Solve as configured at /root/reference/CeresBundleAdjustment.cpp:50-61
//     (HuberLoss(1.0) on every block, SPARSE_SCHUR, max_num_iterations = cfg, everything else Ceres >=1.13 defaults)
//     from the published algorithm: trust-region Levenberg–Marquardt, Jacobi column scaling computed
//     once at the initial point (1/(1+||col||)), LM diagonal clamp(diag(J'J),1e-6,1e32)/radius, Schur elimination of the
//     3-d point blocks, dense Cholesky of the reduced camera matrix, step acceptance rho>1e-3 with radius update
//     r/=max(1/3,1-(2rho-1)^3) | r/=decrease, decrease*=2, termination on max iterations / function 1e-6 /
//     gradient 1e-10 / parameter 1e-8.  FIXED CHOICES: all sums are sequential in observation order; the reduced system
//     is solved by dense Cholesky (Ceres uses a sparse Cholesky of the same matrix).
#include "orc_api.h"
#include "orc_fast.h"
#include "orc_math.h"
#include <cstring>

namespace orc {

// r[2], Jc[2][6] (d r / d [aa, t']), Jp[2][3] (d r / d X). cam = [angle-axis(R^T), -t], K row-major 3x3.
void projection_residual(const double* cam, const double* X, const double* obs, const double* K, double* r,
                         double* Jc, double* Jp) {
    const double fx = K[0], cx = K[2], fy = K[4], cy = K[5];
    const double q[3] = {X[0] + cam[3], X[1] + cam[4], X[2] + cam[5]};
    double p[3], dpdw[9], Rm[9];
    angle_axis_rotate(cam, q, p, dpdw, Rm);
    const double pz = p[2] * -1.0;
    const double u = p[0] / pz * fx + cx, v = p[1] / pz * fy + cy;
    r[0] = obs[0] - u;
    r[1] = obs[1] - v;
    if (!Jc && !Jp) return;
    // du/dp = fx [1/pz, 0, p0/pz^2] (since d pz / d p2 = -1), dv/dp = fy [0, 1/pz, p1/pz^2]; r = obs - proj
    const double ipz = 1.0 / pz;
    const double du[3] = {fx * ipz, 0.0, fx * p[0] * ipz * ipz};
    const double dv[3] = {0.0, fy * ipz, fy * p[1] * ipz * ipz};
    for (int k = 0; k < 3; k++) {
        const double ju_w = du[0] * dpdw[0 * 3 + k] + du[1] * dpdw[1 * 3 + k] + du[2] * dpdw[2 * 3 + k];
        const double jv_w = dv[0] * dpdw[0 * 3 + k] + dv[1] * dpdw[1 * 3 + k] + dv[2] * dpdw[2 * 3 + k];
        const double ju_q = du[0] * Rm[0 * 3 + k] + du[1] * Rm[1 * 3 + k] + du[2] * Rm[2 * 3 + k];
        const double jv_q = dv[0] * Rm[0 * 3 + k] + dv[1] * Rm[1 * 3 + k] + dv[2] * Rm[2 * 3 + k];
        if (Jc) { Jc[0 * 6 + k] = -ju_w; Jc[1 * 6 + k] = -jv_w; Jc[0 * 6 + 3 + k] = -ju_q; Jc[1 * 6 + 3 + k] = -jv_q; }
        if (Jp) { Jp[0 * 3 + k] = -ju_q; Jp[1 * 3 + k] = -jv_q; }
    }
}

struct BAProblem {
    int nc, np, nobs;
    const double* obs; const int* cam_idx; const int* pt_idx; const double* K;
    double huber;
};

// ceres::HuberLoss(a): rho(s) for s = ||r||^2
static inline void huber_rho(double s, double a, double rho[3]) {
    const double b = a * a;
    if (s > b) {
        const double r = std::sqrt(s);
        rho[0] = 2 * a * r - b;
        rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
        rho[2] = -rho[1] / (2 * s);
    } else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
}

// Optional worker threads for the residual/Jacobian evaluation (bench.py's cpu_baseline leg: CeresBundleAdjustment.cpp:58
// num_threads = 4). Observations are independent; the per-observation cost terms are added in observation order afterwards, so
// the result does not depend on the thread count.
// (thread-local: set by the thread that calls ba_solve, so several oracle pipelines can run side by side in one process)
static thread_local Pool* g_ba_pool = nullptr;
static thread_local int g_ba_threads = 1;
void ba_set_pool(Pool* pool, int threads) { g_ba_pool = pool; g_ba_threads = pool ? std::max(1, threads) : 1; }

// cost = 1/2 sum rho(||r||^2); optionally corrected residuals (2*nobs) and Jacobians (nobs*18: Jc 12, Jp 6)
static double evaluate(const BAProblem& P, const double* x, double* res, double* J) {
    const double* cams = x; const double* pts = x + 6 * P.nc;
    std::vector<double> term(P.nobs);
    auto range = [&](int lo, int hi) {
        for (int i = lo; i < hi; i++) {
            double r[2], Jc[12], Jp[6];
            projection_residual(cams + 6 * P.cam_idx[i], pts + 3 * P.pt_idx[i], P.obs + 2 * i, P.K, r, J ? Jc : nullptr, J ? Jp : nullptr);
            const double s = r[0] * r[0] + r[1] * r[1];
            double rho[3];
            huber_rho(s, P.huber, rho);
            term[i] = 0.5 * rho[0];
            // ceres Corrector: rho'' <= 0 for Huber -> scale residual and Jacobian by sqrt(rho')
            const double sr = std::sqrt(rho[1]);
            if (res) { res[2 * i] = r[0] * sr; res[2 * i + 1] = r[1] * sr; }
            if (J) {
                for (int k = 0; k < 12; k++) J[i * 18 + k] = Jc[k] * sr;
                for (int k = 0; k < 6; k++) J[i * 18 + 12 + k] = Jp[k] * sr;
            }
        }
    };
    if (g_ba_pool && g_ba_threads > 1 && P.nobs >= 256) g_ba_pool->parallel_for(P.nobs, g_ba_threads, range);
    else range(0, P.nobs);
    double cost = 0;
    for (int i = 0; i < P.nobs; i++) cost += term[i];
    return cost;
}


int ba_solve(double* cams, int nc, double* pts, int np, const double* obs, const int* cam_idx, const int* pt_idx,
             int nobs, const double* K, double huber, int max_iterations, BASummary* sum) {
    BAProblem P{nc, np, nobs, obs, cam_idx, pt_idx, K, huber};
    const int n = 6 * nc + 3 * np;
    std::vector<double> x(n), cand(n), res(2 * nobs), J((size_t)nobs * 18), scale(n), g(n), diag(n), D2(n), step(n), delta(n);
    memcpy(x.data(), cams, sizeof(double) * 6 * nc);
    memcpy(x.data() + 6 * nc, pts, sizeof(double) * 3 * np);
    // point -> observation lists (observation order)
    std::vector<std::vector<int>> pobs(np);
    for (int i = 0; i < nobs; i++) pobs[pt_idx[i]].push_back(i);

    auto col = [&](int i, int k) -> int { return k < 6 ? 6 * cam_idx[i] + k : 6 * nc + 3 * pt_idx[i] + (k - 6); };
    auto scale_J = [&]() {
        for (int i = 0; i < nobs; i++)
            for (int rr = 0; rr < 2; rr++) {
                for (int k = 0; k < 6; k++) J[i * 18 + rr * 6 + k] *= scale[col(i, k)];
                for (int k = 0; k < 3; k++) J[i * 18 + 12 + rr * 3 + k] *= scale[col(i, 6 + k)];
            }
    };
    auto col_sqnorm = [&](std::vector<double>& out) {
        std::fill(out.begin(), out.end(), 0.0);
        for (int i = 0; i < nobs; i++)
            for (int rr = 0; rr < 2; rr++) {
                for (int k = 0; k < 6; k++) { const double v = J[i * 18 + rr * 6 + k]; out[col(i, k)] += v * v; }
                for (int k = 0; k < 3; k++) { const double v = J[i * 18 + 12 + rr * 3 + k]; out[col(i, 6 + k)] += v * v; }
            }
    };
    auto gradient = [&]() {   // g = J^T r in the (scaled) space of J
        std::fill(g.begin(), g.end(), 0.0);
        for (int i = 0; i < nobs; i++)
            for (int rr = 0; rr < 2; rr++) {
                const double rv = res[2 * i + rr];
                for (int k = 0; k < 6; k++) g[col(i, k)] += J[i * 18 + rr * 6 + k] * rv;
                for (int k = 0; k < 3; k++) g[col(i, 6 + k)] += J[i * 18 + 12 + rr * 3 + k] * rv;
            }
    };

    double x_cost = evaluate(P, x.data(), res.data(), J.data());
    col_sqnorm(diag);
    for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + std::sqrt(diag[i]));   // jacobi_scaling, once
    // gradient of the unscaled problem for the gradient tolerance test
    gradient();
    double gmax = 0;
    for (int i = 0; i < n; i++) gmax = std::max(gmax, std::fabs(g[i]));
    scale_J();
    gradient();
    double x_norm = 0;
    for (int i = 0; i < n; i++) x_norm += x[i] * x[i];
    x_norm = std::sqrt(x_norm);

    sum->initial_cost = x_cost; sum->iterations = 0; sum->successful_steps = 0; sum->termination = 0;
    double radius = 1e4, decrease = 2.0;
    bool reuse_diag = false;
    int invalid = 0;
    const int m = 6 * nc;
    std::vector<double> S((size_t)m * m), rhs(m), EtEinv((size_t)np * 9), gp((size_t)np * 3);

    int iter = 0;
    while (true) {
        if (iter >= max_iterations) { sum->termination = 0; break; }
        if (gmax <= 1e-10) { sum->termination = 2; break; }
        if (radius < 1e-32) { sum->termination = 4; break; }
        iter++;
        // ---- LevenbergMarquardtStrategy::ComputeStep
        if (!reuse_diag) {
            col_sqnorm(diag);
            for (int i = 0; i < n; i++) diag[i] = std::min(std::max(diag[i], 1e-6), 1e32);
        }
        for (int i = 0; i < n; i++) D2[i] = diag[i] / radius;   // lm_diagonal^2
        reuse_diag = true;
        // ---- Schur complement: eliminate points. System (J'J + D2) y = J' r, step = -y
        std::fill(S.begin(), S.end(), 0.0);
        std::fill(rhs.begin(), rhs.end(), 0.0);
        for (int i = 0; i < nobs; i++) {
            const int c = cam_idx[i];
            const double* Jc = &J[i * 18];
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) S[(size_t)(6 * c + a) * m + 6 * c + b] += Jc[a] * Jc[b] + Jc[6 + a] * Jc[6 + b];
                rhs[6 * c + a] += Jc[a] * res[2 * i] + Jc[6 + a] * res[2 * i + 1];
            }
        }
        for (int i = 0; i < m; i++) S[(size_t)i * m + i] += D2[i];
        bool ok = true;
        for (int p = 0; p < np; p++) {
            double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gv[3] = {0, 0, 0};
            for (int i : pobs[p]) {
                const double* Jp = &J[i * 18 + 12];
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) E[a * 3 + b] += Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b];
                    gv[a] += Jp[a] * res[2 * i] + Jp[3 + a] * res[2 * i + 1];
                }
            }
            for (int a = 0; a < 3; a++) E[a * 3 + a] += D2[6 * nc + 3 * p + a];
            // inverse of the SPD 3x3 block via Cholesky
            double L[9];
            memcpy(L, E, sizeof(L));
            if (!cholesky(L, 3)) { ok = false; break; }
            double* Ei = &EtEinv[(size_t)p * 9];
            for (int cI = 0; cI < 3; cI++) {
                double e[3] = {0, 0, 0};
                e[cI] = 1.0;
                cholesky_solve(L, 3, e);
                for (int a = 0; a < 3; a++) Ei[a * 3 + cI] = e[a];
            }
            for (int a = 0; a < 3; a++) gp[(size_t)p * 3 + a] = gv[a];
            // W_i = Jc_i^T Jp_i (6x3), Y_i = W_i Einv
            const int k = (int)pobs[p].size();
            std::vector<double> Wm((size_t)k * 18), Ym((size_t)k * 18);
            for (int t = 0; t < k; t++) {
                const int i = pobs[p][t];
                const double* Jc = &J[i * 18];
                const double* Jp = &J[i * 18 + 12];
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 3; b++) Wm[t * 18 + a * 3 + b] = Jc[a] * Jp[b] + Jc[6 + a] * Jp[3 + b];
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 3; b++)
                        Ym[t * 18 + a * 3 + b] = Wm[t * 18 + a * 3] * Ei[0 * 3 + b] + Wm[t * 18 + a * 3 + 1] * Ei[1 * 3 + b] + Wm[t * 18 + a * 3 + 2] * Ei[2 * 3 + b];
            }
            for (int t = 0; t < k; t++) {
                const int ci = cam_idx[pobs[p][t]];
                for (int a = 0; a < 6; a++)
                    rhs[6 * ci + a] -= Ym[t * 18 + a * 3] * gv[0] + Ym[t * 18 + a * 3 + 1] * gv[1] + Ym[t * 18 + a * 3 + 2] * gv[2];
                for (int u = 0; u < k; u++) {
                    const int cj = cam_idx[pobs[p][u]];
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++)
                            S[(size_t)(6 * ci + a) * m + 6 * cj + b] -=
                                Ym[t * 18 + a * 3] * Wm[u * 18 + b * 3] + Ym[t * 18 + a * 3 + 1] * Wm[u * 18 + b * 3 + 1] + Ym[t * 18 + a * 3 + 2] * Wm[u * 18 + b * 3 + 2];
                }
            }
        }
        std::vector<double> y(n, 0.0);
        if (ok) {
            std::vector<double> Lc(S);
            ok = cholesky(Lc.data(), m);
            if (ok) {
                for (int i = 0; i < m; i++) y[i] = rhs[i];
                cholesky_solve(Lc.data(), m, y.data());
                for (int p = 0; p < np; p++) {
                    double t3[3] = {gp[(size_t)p * 3], gp[(size_t)p * 3 + 1], gp[(size_t)p * 3 + 2]};
                    for (int i : pobs[p]) {
                        const int c = cam_idx[i];
                        const double* Jc = &J[i * 18];
                        const double* Jp = &J[i * 18 + 12];
                        // W^T y_c = Jp^T (Jc y_c)
                        double jy0 = 0, jy1 = 0;
                        for (int a = 0; a < 6; a++) { jy0 += Jc[a] * y[6 * c + a]; jy1 += Jc[6 + a] * y[6 * c + a]; }
                        for (int a = 0; a < 3; a++) t3[a] -= Jp[a] * jy0 + Jp[3 + a] * jy1;
                    }
                    const double* Ei = &EtEinv[(size_t)p * 9];
                    for (int a = 0; a < 3; a++) y[6 * nc + 3 * p + a] = Ei[a * 3] * t3[0] + Ei[a * 3 + 1] * t3[1] + Ei[a * 3 + 2] * t3[2];
                }
            }
        }
        double model_change = 0;
        if (ok) {
            for (int i = 0; i < n; i++) step[i] = -y[i];
            for (int i = 0; i < nobs; i++) {
                for (int rr = 0; rr < 2; rr++) {
                    double mr = 0;
                    for (int k = 0; k < 6; k++) mr += J[i * 18 + rr * 6 + k] * step[col(i, k)];
                    for (int k = 0; k < 3; k++) mr += J[i * 18 + 12 + rr * 3 + k] * step[col(i, 6 + k)];
                    model_change -= mr * (res[2 * i + rr] + mr / 2.0);
                }
            }
        }
        if (!ok || !(model_change > 0.0)) {
            // HandleInvalidStep
            if (++invalid >= 5) { sum->termination = 4; break; }
            radius /= decrease; decrease *= 2; reuse_diag = true;
            continue;
        }
        invalid = 0;
        double step_norm = 0;
        for (int i = 0; i < n; i++) { delta[i] = step[i] * scale[i]; cand[i] = x[i] + delta[i]; step_norm += delta[i] * delta[i]; }
        step_norm = std::sqrt(step_norm);
        const double cand_cost = evaluate(P, cand.data(), nullptr, nullptr);
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum->termination = 3; break; }
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= 1e-6 * x_cost) { sum->termination = 1; break; }
        const double rel = cost_change / model_change;
        if (rel > 1e-3) {
            x = cand;
            x_norm = 0;
            for (int i = 0; i < n; i++) x_norm += x[i] * x[i];
            x_norm = std::sqrt(x_norm);
            x_cost = evaluate(P, x.data(), res.data(), J.data());
            gradient();
            gmax = 0;
            for (int i = 0; i < n; i++) gmax = std::max(gmax, std::fabs(g[i]));
            scale_J();
            sum->successful_steps++;
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
            radius = std::min(1e16, radius);
            decrease = 2.0;
            reuse_diag = false;
        } else {
            radius /= decrease; decrease *= 2; reuse_diag = true;
        }
    }
    sum->iterations = iter;
    sum->final_cost = x_cost;
    memcpy(cams, x.data(), sizeof(double) * 6 * nc);
    memcpy(pts, x.data() + 6 * nc, sizeof(double) * 3 * np);
    return 0;
}

}  // namespace orc

extern "C" {
void orc_projection_residual(const double* cam, const double* X, const double* obs, const double* K, double* r,
                             double* Jc, double* Jp) {
    orc::projection_residual(cam, X, obs, K, r, Jc, Jp);
}
void orc_ba_residuals(const double* cams, int nc, const double* pts, int np, const double* obs, const int* cam_idx,
                      const int* pt_idx, int nobs, const double* K, double* out_r, double* out_J) {
    (void)nc; (void)np;
    for (int i = 0; i < nobs; i++) {
        double Jc[12], Jp[6];
        orc::projection_residual(cams + 6 * cam_idx[i], pts + 3 * pt_idx[i], obs + 2 * i, K, out_r + 2 * i, Jc, Jp);
        for (int rr = 0; rr < 2; rr++) {
            for (int k = 0; k < 6; k++) out_J[i * 18 + rr * 9 + k] = Jc[rr * 6 + k];
            for (int k = 0; k < 3; k++) out_J[i * 18 + rr * 9 + 6 + k] = Jp[rr * 3 + k];
        }
    }
}
int orc_ba_solve(double* cams, int nc, double* pts, int np, const double* obs, const int* cam_idx, const int* pt_idx,
                 int nobs, const double* K, double huber, int max_iterations, double* summary5) {
    orc::BASummary s;
    int rc = orc::ba_solve(cams, nc, pts, np, obs, cam_idx, pt_idx, nobs, K, huber, max_iterations, &s);
    summary5[0] = s.initial_cost; summary5[1] = s.final_cost; summary5[2] = s.iterations; summary5[3] = s.successful_steps; summary5[4] = s.termination;
    return rc;
}
}
