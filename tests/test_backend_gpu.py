"""GPU parity (through the C ABI) of the back-end: ProjectionResidual + Jacobians, the device-resident LM/Schur bundle
adjustment, and RANSAC/EPnP + LM refit, against the oracle on seeded problems.
Tolerances (float64 paths whose summation order differs between the CPU loops and the GPU reduction trees / MFMA):
  residuals/Jacobians 1e-9 relative; BA parameters 1e-6 relative after 5 LM iterations; PnP pose 1e-6; inlier sets exact."""
import numpy as np
import pytest

import orc_binding as ob
import scenes

pytestmark = pytest.mark.gpu


def _ctx(gpu_ctx_factory, **kw):
    return gpu_ctx_factory(64, 64, n_slots=1, max_tracks=4096, **kw)


@pytest.mark.parametrize("seed,nc,npts", [(1, 5, 300), (2, 3, 120), (3, 10, 800)])
def test_ba_residuals_match_oracle(gpu_ctx_factory, seed, nc, npts):
    P = scenes.ba_problem(seed, nc=nc, npts=npts)
    ctx = _ctx(gpu_ctx_factory)
    r, J = ctx.ba_residuals(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K)
    rr, JJ = ob.ba_residuals(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K)
    np.testing.assert_allclose(r, rr, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(J, JJ, rtol=1e-9, atol=1e-9)
    # KA: residual equals the numpy twin of ProjectionResidual.h
    for i in range(0, len(r), 97):
        ref = P["obs"][i] - scenes.project_ref(P["cams"][P["cam_idx"][i]], P["pts"][P["pt_idx"][i]])
        np.testing.assert_allclose(r[i], ref, rtol=1e-10, atol=1e-9)


def test_ba_residual_small_angle_branch(gpu_ctx_factory):
    ctx = _ctx(gpu_ctx_factory)
    cams = np.array([[0, 0, 0, 0.1, 0.2, 0.3], [1e-9, -2e-9, 1e-9, 0, 0, 0], [2.5, 0.3, -1.0, 0.5, 0.1, -0.2]])
    pts = np.array([[1.0, -0.5, -12.0], [-3.0, 1.0, -20.0]])
    ci = np.array([0, 1, 2, 0, 1, 2], np.int32)
    pi = np.array([0, 0, 0, 1, 1, 1], np.int32)
    obs = np.tile(np.array([500.0, 200.0]), (6, 1))
    r, J = ctx.ba_residuals(cams, pts, obs, ci, pi, scenes.K)
    rr, JJ = ob.ba_residuals(cams, pts, obs, ci, pi, scenes.K)
    np.testing.assert_allclose(r, rr, rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(J, JJ, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("seed,nc,npts,iters", [(1, 5, 300, 5), (4, 5, 549, 5), (5, 3, 200, 5), (6, 10, 1000, 5), (7, 5, 300, 50),
                                                (8, 20, 1500, 5)])
def test_ba_solve_matches_oracle(gpu_ctx_factory, seed, nc, npts, iters):
    P = scenes.ba_problem(seed, nc=nc, npts=npts)
    ctx = _ctx(gpu_ctx_factory)
    cams, pts, s = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, iters)
    rc, rp, rs = ob.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, iters)
    np.testing.assert_allclose(s.initial_cost, rs["initial_cost"], rtol=1e-12)
    if iters > 5:
        # 50 LM iterations of a gauge-free problem: accept/reject decisions near rho = 1e-3 amplify 1e-16 differences, so
        # only the reached cost level is comparable (the metric config runs 5 iterations)
        assert s.iterations == rs["iterations"]
        np.testing.assert_allclose(s.final_cost, rs["final_cost"], rtol=1e-3)
        return
    assert s.iterations == rs["iterations"] and s.successful_steps == rs["successful_steps"] and s.termination == rs["termination"]
    np.testing.assert_allclose(s.final_cost, rs["final_cost"], rtol=1e-8)
    np.testing.assert_allclose(cams, rc, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pts, rp, rtol=1e-6, atol=1e-6)
    assert s.final_cost < 0.5 * s.initial_cost
    # reference write-back is float32 (Feature3D::update): identical after rounding except for rare 1-ulp ties
    assert (pts.astype(np.float32) != rp.astype(np.float32)).mean() < 0.03


def test_ba_solve_is_bitwise_reproducible(gpu_ctx_factory):
    P = scenes.ba_problem(11, nc=5, npts=400)
    ctx = _ctx(gpu_ctx_factory)
    a = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K)
    b = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_ba_noiseless_scene_converges_to_zero_cost(gpu_ctx_factory):
    """KA8: exact observations, perturbed start -> cost collapses (gauge-free quantities only)."""
    P = scenes.ba_problem(21, nc=5, npts=300, noise=0.0, outlier_every=0)
    obs = np.array([scenes.project_ref(P["cams_true"][c], P["pts_true"][p]) for c, p in zip(P["cam_idx"], P["pt_idx"])])
    ctx = _ctx(gpu_ctx_factory)
    cams, pts, s = ctx.ba_solve(P["cams"], P["pts"], obs, P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 50)
    assert s.final_cost < 1e-6 * s.initial_cost


@pytest.mark.parametrize("seed,m,frac", [(1, 400, 0.2), (2, 150, 0.1), (3, 549, 0.35), (4, 1099, 0.2), (5, 60, 0.0)])
def test_pnp_ransac_matches_oracle(gpu_ctx_factory, seed, m, frac):
    P = scenes.pnp_problem(seed, m=m, outlier_frac=frac)
    ctx = _ctx(gpu_ctx_factory)
    guess_r, guess_t = np.array([0.3, -0.2, 0.1]), np.array([1.0, 2.0, -30.0])   # the reference passes the ABSOLUTE pose (Q8)
    rv, tv, inl = ctx.pnp_ransac(P["obj"], P["img"], scenes.K, guess_r, guess_t)
    rr, rt, rinl, hyp = ob.pnp_ransac(P["obj"], P["img"], scenes.K, guess_r, guess_t)
    assert np.array_equal(inl, rinl), "inlier index list differs"
    np.testing.assert_allclose(rv, rr, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(tv, rt, rtol=1e-6, atol=1e-8)
    # pose is close to the truth and no gross outlier is kept
    assert np.abs(rv - P["rvec_true"]).max() < 5e-3 and np.abs(tv - P["tvec_true"]).max() < 5e-2
    assert P["outliers"][inl].mean() < 0.05


def test_pnp_degenerate_inputs(pmv, gpu_ctx_factory):
    ctx = _ctx(gpu_ctx_factory)
    P = scenes.pnp_problem(9, m=5)
    with pytest.raises(pmv.PmvError) as e:
        ctx.pnp_ransac(P["obj"], P["img"], scenes.K, np.zeros(3), np.zeros(3))
    assert e.value.code == -5
    # all-outlier data: no model with more than 4 inliers -> empty inlier list, pose = last evaluated model
    rng = np.random.default_rng(0)
    obj = rng.uniform(-5, 5, (50, 3)).astype(np.float32) + [0, 0, 20]
    img = rng.uniform(0, 1200, (50, 2)).astype(np.float32)
    rv, tv, inl = ctx.pnp_ransac(obj, img, scenes.K, np.zeros(3), np.zeros(3))
    rr, rt, rinl, _ = ob.pnp_ransac(obj, img, scenes.K, np.zeros(3), np.zeros(3))
    assert len(inl) == len(rinl)


@pytest.mark.parametrize("seed,n", [(1, 400), (2, 63), (3, 1), (4, 1500)])
def test_triangulate_candidates_bit_exact(gpu_ctx_factory, seed, n):
    """recoverPose's DLT + cheirality on the GPU against the host loops (same operation order): bit-exact."""
    P = scenes.two_view_problem(seed, n=n)
    ctx = _ctx(gpu_ctx_factory)
    Q, mask, good = ctx.triangulate_candidates(P["q1"], P["q2"], P["P1x4"], P["mask"])
    rQ, rmask, rgood = ob.triangulate_candidates(P["q1"], P["q2"], P["P1x4"], P["mask"])
    assert np.array_equal(Q, rQ)
    assert np.array_equal(mask, rmask) and np.array_equal(good, rgood)
    # KA: the true candidate wins the cheirality vote and its points reproduce the scene up to the noise
    assert good[0] == max(good) and good[0] >= 0.9 * P["mask"].sum()
    X = (Q[0, :3] / Q[0, 3]).T
    ok = mask[0].astype(bool)
    if ok.sum() > 10:
        assert np.median(np.linalg.norm(X[ok] - P["X"][ok], axis=1) / P["X"][ok, 2]) < 0.05


def test_ba_duplicate_point_camera_observations(gpu_ctx_factory):
    """The reference can observe one landmark twice in the same frame (two same-pixel features): both residual blocks
    enter the problem. The point kernel folds such duplicates into the first entry of the (point, camera) group."""
    P = scenes.ba_problem(13, nc=5, npts=250)
    rng = np.random.default_rng(5)
    dup = rng.choice(len(P["obs"]), 60, replace=False)
    obs = np.vstack([P["obs"], P["obs"][dup] + rng.normal(0, 0.4, (60, 2))])
    ci = np.concatenate([P["cam_idx"], P["cam_idx"][dup]]).astype(np.int32)
    pi = np.concatenate([P["pt_idx"], P["pt_idx"][dup]]).astype(np.int32)
    ctx = _ctx(gpu_ctx_factory)
    cams, pts, s = ctx.ba_solve(P["cams"], P["pts"], obs, ci, pi, scenes.K, 1.0, 5)
    rc, rp, rs = ob.ba_solve(P["cams"], P["pts"], obs, ci, pi, scenes.K, 1.0, 5)
    np.testing.assert_allclose(s.initial_cost, rs["initial_cost"], rtol=1e-12)
    assert s.iterations == rs["iterations"] and s.successful_steps == rs["successful_steps"]
    np.testing.assert_allclose(s.final_cost, rs["final_cost"], rtol=1e-8)
    np.testing.assert_allclose(cams, rc, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pts, rp, rtol=1e-6, atol=1e-6)


def test_backend_argument_errors(pmv, gpu_ctx_factory):
    ctx = gpu_ctx_factory(64, 64, n_slots=1, max_tracks=128, max_ba_cams=4, max_ba_points=64, max_ba_obs=256)
    P = scenes.ba_problem(2, nc=3, npts=40)
    c0, p0, s0 = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 0)   # 0 iterations: untouched
    assert np.array_equal(c0, P["cams"]) and np.array_equal(p0, P["pts"]) and s0.iterations == 0
    with pytest.raises(pmv.PmvError) as e:     # LM iteration count must be 0..512
        ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 513)
    assert e.value.code == -2
    with pytest.raises(pmv.PmvError) as e:     # more cameras than the context was created for
        Q = scenes.ba_problem(2, nc=5, npts=40)
        ctx.ba_solve(Q["cams"], Q["pts"], Q["obs"], Q["cam_idx"], Q["pt_idx"], scenes.K)
    assert e.value.code == -3
    T = scenes.two_view_problem(1, n=200)
    with pytest.raises(pmv.PmvError) as e:     # more correspondences than max_tracks
        ctx.triangulate_candidates(T["q1"], T["q2"], T["P1x4"], T["mask"])
    assert e.value.code == -3
    bad = P["cam_idx"].copy(); bad[3] = 7
    with pytest.raises(pmv.PmvError) as e:     # index out of range is caught on the host, never reaches a kernel
        ctx.ba_solve(P["cams"], P["pts"], P["obs"], bad, P["pt_idx"], scenes.K)
    assert e.value.code == -2


def test_ba_single_workgroup_variant_agrees(gpu_ctx_factory):
    """PMV_BA_MODE=single selects the one-workgroup persistent LM kernel (k_ba_lm, the first implementation, kept as an A/B
    reference); the mode is read once per process, so it runs in a child process. Same problem, same answer within the BA bar."""
    import json, os, subprocess, sys
    P = scenes.ba_problem(17, nc=5, npts=300)
    ctx = _ctx(gpu_ctx_factory)
    cams, pts, s = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    code = ("import sys, json, importlib, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); import scenes; "
            "pmv = importlib.import_module('practical-multi-view_amd'); ctx = pmv.Context(64, 64, n_slots=1); "
            "P = scenes.ba_problem(17, nc=5, npts=300); "
            "c, p, s = ctx.ba_solve(P['cams'], P['pts'], P['obs'], P['cam_idx'], P['pt_idx'], scenes.K, 1.0, 5); "
            "print(json.dumps(dict(cams=c.tolist(), pts=p.tolist(), cost=s.final_cost, it=s.iterations, ok=s.successful_steps)))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, PMV_BA_MODE="single"), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["it"] == s.iterations and r["ok"] == s.successful_steps
    np.testing.assert_allclose(r["cost"], s.final_cost, rtol=1e-8)
    np.testing.assert_allclose(np.array(r["cams"]), cams, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(np.array(r["pts"]), pts, rtol=1e-6, atol=1e-6)


def test_ba_mode_one_workgroup_matches_oracle_and_default(gpu_ctx_factory):
    """pmv_set_ba_mode(ctx, 1): the whole LM loop of a call in ONE workgroup (k_ba_lm; the batched engine's k_ba_lm_batch runs the
    same body per problem). Same oracle bars as the default chain of launches, switchable per context and back."""
    P = scenes.ba_problem(23, nc=5, npts=400)
    ctx = _ctx(gpu_ctx_factory)
    rc, rp, rs = ob.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    d_cams, d_pts, d_s = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    ctx.set_ba_mode(1)
    cams, pts, s = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    again = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    assert np.array_equal(cams, again[0]) and np.array_equal(pts, again[1]), "mode 1 is not bitwise reproducible"
    assert s.iterations == rs["iterations"] and s.successful_steps == rs["successful_steps"] and s.termination == rs["termination"]
    np.testing.assert_allclose(s.initial_cost, rs["initial_cost"], rtol=1e-12)
    np.testing.assert_allclose(s.final_cost, rs["final_cost"], rtol=1e-8)
    np.testing.assert_allclose(cams, rc, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pts, rp, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cams, d_cams, rtol=1e-6, atol=1e-8)
    ctx.set_ba_mode(0)
    back = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    assert np.array_equal(back[0], d_cams) and np.array_equal(back[1], d_pts), "switching back does not restore the default path"
    with pytest.raises(Exception):
        ctx.set_ba_mode(7)


def test_fivepoint_round_on_device_is_bit_exact(pmv, orc, gpu_ctx_factory):
    """the hypothesis half of cv::findEssentialMat on the GPU (k_fivepoint_hyp: Nister's solver, one thread per sample;
    k_fivepoint_score: float32 Sampson inlier counts) against the host solver that the CPU known-answer tests pin
    (tests/test_twoview_host.py): every essential matrix bit-exact, every inlier count exact, degenerate samples give no model."""
    import ctypes as C
    rng = np.random.default_rng(12)
    n = 400
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 2, n), rng.uniform(5, 40, n)], 1)
    rv = rng.normal(0, 0.03, 3); th = np.linalg.norm(rv); k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.08, -0.03, -1.0]); t /= np.linalg.norm(t)
    Xc = X @ R.T + t
    f = 718.856
    q1 = np.floor(X[:, :2] / X[:, 2:3] * f + rng.normal(0, 0.3, (n, 2))) / f     # integer pixels like the reference's cv::Point (Q12)
    q2 = np.floor(Xc[:, :2] / Xc[:, 2:3] * f + rng.normal(0, 0.3, (n, 2))) / f
    q2[::7] += rng.normal(0, 0.05, q2[::7].shape)                                 # outliers
    nh = 32
    samples = np.zeros((nh, 5), np.int32)
    orc.lib.orc_host_five_point_samples(n, nh, samples.ctypes.data_as(C.POINTER(C.c_int)))
    samples[5] = samples[5][[0, 0, 1, 2, 3]]                                      # a repeated index: rank-deficient sample
    thr = np.float32((1.0 / f) ** 2)
    ctx = gpu_ctx_factory(640, 480, n_slots=1, max_tracks=1024)
    models, nm, counts = ctx.fivepoint_hypotheses(q1, q2, samples, thr)
    f64p = C.POINTER(C.c_double)
    x1 = np.concatenate([q1, np.ones((n, 1))], 1); x2 = np.concatenate([q2, np.ones((n, 1))], 1)
    total = 0
    for h in range(nh):
        Es = np.zeros(90)
        s1 = np.ascontiguousarray(q1[samples[h]]); s2 = np.ascontiguousarray(q2[samples[h]])
        want = orc.lib.orc_host_five_point(s1.ctypes.data_as(f64p), s2.ctypes.data_as(f64p), Es.ctypes.data_as(f64p))
        assert nm[h] == want, f"hypothesis {h}: {nm[h]} models on the device, {want} on the host"
        assert np.array_equal(models[h].reshape(90)[: 9 * want], Es[: 9 * want]), f"hypothesis {h}: models are not bit-exact"
        for mi in range(want):
            E = Es[9 * mi: 9 * mi + 9].reshape(3, 3)
            Ex1 = x1 @ E.T; Etx2 = x2 @ E
            err = ((x2 * Ex1).sum(1) ** 2 / (Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2)).astype(np.float32)
            # (the count is a threshold decision on float32 values of a float64 expression: numpy's evaluation order of the sums may
            # differ in the last bit, so points within 1e-6 relative of the threshold are excluded from the comparison)
            sure_in = int((err <= thr * np.float32(1 - 1e-6)).sum()); sure_out = int((err > thr * np.float32(1 + 1e-6)).sum())
            assert sure_in <= counts[h, mi] <= n - sure_out
        total += want
    assert nm[5] == 0 and total > 40 and counts.max() > 0.7 * n


def test_pipeline_with_device_fivepoint_equals_host_fivepoint(pmv, gpu_ctx_factory):
    """the triangulator's RANSAC hypotheses on the GPU (device_fivepoint = 1) or on host threads: identical features and poses,
    for one sequence and through the batch engine"""
    cfg = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
    n = 45
    frames, poses = pmv.synth_sequence(1002, 0, n, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=16)
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=2 * n, max_tracks=4096)
    ctx.frames_stage(0, frames); ctx.frames_stage(n, frames)
    a = ctx.pipeline_run(n, cfg["w"], cfg["h"], K, poses, threaded=1, n_threads=4)
    b = ctx.pipeline_run(n, cfg["w"], cfg["h"], K, poses, threaded=1, device_fivepoint=1)
    assert a.stats["tri_calls"] >= 3
    assert np.array_equal(a.poses, b.poses)
    for x, y in zip(a.features, b.features):
        assert np.array_equal(x, y)
    for r in ctx.pipeline_run_batch([(0, n, poses), (n, n, poses)], cfg["w"], cfg["h"], K, device_fivepoint=1):
        assert np.array_equal(a.poses, r.poses)
