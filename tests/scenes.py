"""Seeded synthetic back-end problems (BA windows, PnP correspondences) shared by CPU and GPU tests."""
import numpy as np

K = np.array([707.0912, 0, 601.8873, 0, 707.0912, 183.1104, 0, 0, 1.0])


def project_ref(cam, X):
    """numpy twin of ProjectionResidual.h:38-58 (projection part): cam = [angle-axis, t']"""
    q = X + cam[3:]
    a = cam[:3]
    th2 = a @ a
    if th2 > np.finfo(float).eps:
        th = np.sqrt(th2)
        w = a / th
        p = q * np.cos(th) + np.cross(w, q) * np.sin(th) + w * (w @ q) * (1 - np.cos(th))
    else:
        p = q + np.cross(a, q)
    pz = -p[2]
    return np.array([p[0] / pz * K[0] + K[2], p[1] / pz * K[4] + K[5]])


def ba_problem(seed, nc=5, npts=300, vis=0.8, noise=0.3, outlier_every=37, perturb=True):
    rng = np.random.default_rng(seed)
    cams_true = np.zeros((nc, 6))
    for i in range(nc):
        cams_true[i] = [0.01 * i, -0.02 * i, 0.005 * i, 0.1 * i, 0.02 * i, 0.9 * i]
    pts_true = np.stack([rng.uniform(-8, 8, npts), rng.uniform(-3, 2, npts), rng.uniform(-45 - 0.9 * nc, -8 - 0.9 * nc, npts)], 1)
    obs, ci, pi = [], [], []
    for c in range(nc):
        for p in range(npts):
            if rng.random() < vis:
                obs.append(project_ref(cams_true[c], pts_true[p]) + rng.normal(0, noise, 2))
                ci.append(c)
                pi.append(p)
    obs = np.array(obs)
    if outlier_every:
        obs[::outlier_every] += rng.normal(0, 30, obs[::outlier_every].shape)
    # every point must be observed at least once
    seen = np.zeros(npts, bool)
    seen[pi] = True
    remap = -np.ones(npts, int)
    remap[seen] = np.arange(seen.sum())
    pi = remap[np.array(pi)]
    pts_true = pts_true[seen]
    cams = cams_true.copy()
    pts = pts_true.copy()
    if perturb:
        cams = cams + rng.normal(0, 0.002, cams.shape)
        cams[:, 3:] += rng.normal(0, 0.05, (nc, 3))
        pts = pts + rng.normal(0, 0.2, pts.shape)
    # integer-valued observations like the reference's Feature coordinates
    obs = np.floor(obs)
    return dict(cams=cams, pts=pts, obs=obs, cam_idx=np.array(ci, np.int32), pt_idx=pi.astype(np.int32),
                cams_true=cams_true, pts_true=pts_true)


def pnp_problem(seed, m=400, outlier_frac=0.2, noise=0.5):
    rng = np.random.default_rng(seed)
    rtrue = np.array([0.01, -0.03, 0.005]) + rng.normal(0, 0.005, 3)
    ttrue = np.array([0.05, -0.02, -0.9]) + rng.normal(0, 0.05, 3)
    X = np.stack([rng.uniform(-8, 8, m), rng.uniform(-3, 2, m), rng.uniform(6, 40, m)], 1).astype(np.float32)
    th = np.linalg.norm(rtrue)
    k = rtrue / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    Xc = (R @ X.T).T + ttrue
    uv = np.stack([Xc[:, 0] / Xc[:, 2] * K[0] + K[2], Xc[:, 1] / Xc[:, 2] * K[4] + K[5]], 1)
    uv = np.floor(uv + rng.normal(0, noise, uv.shape)).astype(np.float32)
    out = rng.random(m) < outlier_frac
    uv[out] += rng.normal(0, 40, (int(out.sum()), 2)).astype(np.float32)
    return dict(obj=X, img=uv, rvec_true=rtrue, tvec_true=ttrue, outliers=out)


def two_view_problem(seed, n=400, outlier_frac=0.15, noise=0.3):
    """Two views of a point cloud in front of camera 0 = [I|0]; camera 1 = [R|t] with |t| = 1 (essential-matrix scale).
    Returns normalised image points q1, q2 (n,2), the four candidate matrices of decomposeEssentialMat (true one first:
    [R|t], [R2|t], [R|-t], [R2|-t]) as (4,3,4), the RANSAC-style inlier mask and the true points."""
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 2, n), rng.uniform(5, 40, n)], 1)
    rv = rng.normal(0, 0.02, 3)
    th = np.linalg.norm(rv); k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = np.array([0.05, -0.02, -1.0]); t /= np.linalg.norm(t)
    Xc = (R @ X.T).T + t
    q1 = X[:, :2] / X[:, 2:3]
    q2 = Xc[:, :2] / Xc[:, 2:3]
    q1 = q1 + rng.normal(0, noise / K[0], q1.shape)
    q2 = q2 + rng.normal(0, noise / K[0], q2.shape)
    mask = (rng.random(n) >= outlier_frac).astype(np.uint8)
    # the "twisted" rotation of decomposeEssentialMat: R2 = R_t(pi) R
    Rt = 2 * np.outer(t, t) - np.eye(3)
    R2 = Rt @ R
    P = np.stack([np.hstack([R, t[:, None]]), np.hstack([R2, t[:, None]]), np.hstack([R, -t[:, None]]), np.hstack([R2, -t[:, None]])])
    return dict(q1=q1, q2=q2, P1x4=P, mask=mask, X=X)
