"""ctypes binding of oracle/liborc.so — the CPU restatement used ONLY as the checker (tests, smoke, cpu_baseline)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def pyr_down(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.orc_pyr_down(_p(img, _u8p), w, h, _p(out, _u8p))
        return out

    def bgr2gray(self, bgr):
        b = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = b.shape
        out = np.zeros((h, w), np.uint8)
        self.lib.orc_bgr2gray(_p(b, _u8p), w, h, 3 * w, _p(out, _u8p))
        return out

    def scharr(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((h, w, 2), np.int16)
        self.lib.orc_scharr(_p(img, _u8p), w, h, _p(out, _i16p))
        return out

    def lk_track(self, prev, nxt, pts, win=32, max_level=4, max_iter=30, eps=0.01, min_eig=1e-4):
        prev = np.ascontiguousarray(prev, np.uint8)
        nxt = np.ascontiguousarray(nxt, np.uint8)
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = p.shape[0]
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        h, w = prev.shape
        lv = self.lib.orc_lk_track(_p(prev, _u8p), _p(nxt, _u8p), w, h, _p(p, _f32p), n, win, max_level, max_iter,
                                   C.c_double(eps), C.c_float(min_eig), _p(out, _f32p), _p(st, _u8p), _p(err, _f32p))
        return out, st, err, lv

    def lk_track_fast(self, prev, nxt, pts, nthreads=1):
        """the speed-oriented twin (orc_fast.cpp): must equal lk_track bit for bit"""
        prev = np.ascontiguousarray(prev, np.uint8)
        nxt = np.ascontiguousarray(nxt, np.uint8)
        p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = p.shape[0]
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        h, w = prev.shape
        rc = self.lib.orc_lk_track_fast(_p(prev, _u8p), _p(nxt, _u8p), w, h, _p(p, _f32p), n, 32, 4, 30, C.c_double(0.01), C.c_float(1e-4),
                                        _p(out, _f32p), _p(st, _u8p), _p(err, _f32p), nthreads)
        assert rc == 0
        return out, st, err

    def pyr_down_fast(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.orc_pyr_down_fast(_p(img, _u8p), w, h, _p(out, _u8p))
        return out

    def fast9_cell(self, img, cell, max_kp, threshold=10, nonmax=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        x0, y0, cw, ch = [int(v) for v in cell]
        xy = np.zeros((max(max_kp, 1), 2), np.int32)
        rs = np.zeros(max(max_kp, 1), np.float32)
        n = self.lib.orc_fast9_cell(_p(img, _u8p), w, h, x0, y0, cw, ch, threshold, 1 if nonmax else 0, max_kp, _p(xy, _i32p), _p(rs, _f32p))
        return xy[:n].copy(), rs[:n].copy()

    def knn_match(self, src, cmp, src_xy, cmp_xy, neighbours=7, window=15):
        src = np.ascontiguousarray(src, np.uint8)
        cmp = np.ascontiguousarray(cmp, np.uint8)
        h, w = src.shape
        s = np.ascontiguousarray(src_xy, np.int32).reshape(-1, 2)
        c = np.ascontiguousarray(cmp_xy, np.int32).reshape(-1, 2)
        best = np.zeros(max(len(s), 1), np.int32)
        err = np.zeros(max(len(s), 1), np.float32)
        self.lib.orc_knn_match(_p(src, _u8p), _p(cmp, _u8p), w, h, _p(s, _i32p), len(s), _p(c, _i32p), len(c), neighbours, window, _p(best, _i32p), _p(err, _f32p))
        return best[: len(s)].copy(), err[: len(s)].copy()

    def gftt_cell(self, img, cell, max_corners, quality=0.01, min_dist=5.0, want_eig=False):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        x0, y0, cw, ch = [int(v) for v in cell]
        xy = np.zeros((max_corners if max_corners > 0 else 65536, 2), np.int32)   # max_corners <= 0: no limit
        eig = np.zeros((ch, cw), np.float32)
        n = self.lib.orc_gftt_cell(_p(img, _u8p), w, h, x0, y0, cw, ch, max_corners, C.c_double(quality),
                                   C.c_double(min_dist), _p(xy, _i32p), _p(eig, _f32p))
        return (xy[:n].copy(), eig) if want_eig else xy[:n].copy()

    def shitomasi_cell(self, img, cell, max_feats, quality=0.4, want_resp=False):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        x0, y0, cw, ch = [int(v) for v in cell]
        xy = np.zeros((max(max_feats, 1), 2), np.int32)
        sc = np.zeros(max(max_feats, 1), np.float64)
        R = np.zeros((ch, cw), np.float64)
        n = self.lib.orc_shitomasi_cell(_p(img, _u8p), w, h, x0, y0, cw, ch, max_feats, C.c_double(quality),
                                        _p(xy, _i32p), _p(sc, _f64p), _p(R, _f64p))
        return (xy[:n].copy(), sc[:n].copy(), R) if want_resp else (xy[:n].copy(), sc[:n].copy())


_cached = None


def load():
    global _cached
    if _cached is None:
        so = os.path.join(ROOT, "oracle", "liborc.so")
        # make decides (liborc.so depends on oracle/*.cpp, *.h and the host/ sources it shares): a stale library must never be the checker
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        _cached = Oracle(C.CDLL(so))
    return _cached


class PipelineParams(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("w", C.c_int), ("h", C.c_int), ("min_tracked_features", C.c_int),
                ("tracked_features_tol", C.c_int), ("init_frames", C.c_int), ("bundle_size", C.c_int),
                ("ba_iterations", C.c_int), ("extractor", C.c_int), ("threaded", C.c_int), ("n_threads", C.c_int),
                ("reserved", C.c_int), ("matcher", C.c_int)]


class PipelineResult:
    """poses (n,12: R row-major then t), per-frame feature triples (col,row,landmark) in container order, stats"""

    def __init__(self, lib, prefix, handle, want_features=True):
        g = lambda name: getattr(lib, prefix + name)
        n = g("num_poses")(handle)
        self.poses = np.zeros((n, 12), np.float64)
        if n:
            g("get_poses")(handle, _p(self.poses, _f64p))
        nf = g("num_frames")(handle) if want_features else 0
        self.features = []
        for k in range(nf):
            c = g("frame_feature_count")(handle, k)
            a = np.zeros((c, 3), np.int32)
            if c:
                g("get_frame_features")(handle, k, _p(a, _i32p))
            self.features.append(a)
        self.corr_counts = [getattr(lib, prefix + "frame_corr_count")(handle, k) for k in range(nf)] if hasattr(lib, prefix + "frame_corr_count") else None
        st = np.zeros(g("stats_count")(), np.float64)
        g("get_stats")(handle, _p(st, _f64p))
        keys = ["lk_calls", "lk_points", "detect_calls", "pnp_calls", "pnp_points", "tri_calls", "ba_calls", "ba_obs",
                "ba_points", "heuristic_motion", "seconds", "init_offset", "n_landmarks", "scale", "t_lk", "t_detect", "t_pnp", "t_tri",
             "t_ba", "t_pnp_kernel", "t_ba_kernel", "t_tri_essential", "t_tri_pose", "tri_hypotheses", "tri_ahead"]
        self.stats = dict(zip(keys, st[:len(keys)]))


def run_pipeline(frames, K, gt_poses, min_tracked=400, tol=150, init_frames=5, bundle_size=5, ba_iterations=5,
                 extractor=0, threaded=0, n_threads=1, fast=False, lib=None, matcher=0, want_features=True):
    """the oracle pipeline; fast=True: the speed-oriented twins (orc_fast.cpp; identical results), lib: another build of the
    same sources (bench.py times oracle/liborc_fast.so, built -O3 -march=native on the machine it runs on)"""
    o = load()
    lib = lib or o.lib
    lib.orc_pipeline_run.restype = C.c_void_p
    lib.orc_pipeline_frame_corr_count.argtypes = [C.c_void_p, C.c_int]
    for f in ("orc_pipeline_free", "orc_pipeline_num_poses", "orc_pipeline_get_poses", "orc_pipeline_num_frames",
              "orc_pipeline_frame_feature_count", "orc_pipeline_get_frame_features", "orc_pipeline_get_stats"):
        getattr(lib, f).argtypes = [C.c_void_p] + ([C.c_int] if "frame_f" in f or "get_frame" in f else [])
    lib.orc_pipeline_get_poses.argtypes = [C.c_void_p, _f64p]
    lib.orc_pipeline_get_frame_features.argtypes = [C.c_void_p, C.c_int, _i32p]
    lib.orc_pipeline_get_stats.argtypes = [C.c_void_p, _f64p]
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    P = PipelineParams(n, w, h, min_tracked, tol, init_frames, bundle_size, ba_iterations, extractor, threaded,
                       n_threads, 1 if fast else 0, matcher)
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    gt = np.ascontiguousarray(gt_poses, np.float64).reshape(n, 12)
    lib.orc_pipeline_run.argtypes = [C.POINTER(PipelineParams), _u8p, _f64p, _f64p]
    hnd = lib.orc_pipeline_run(C.byref(P), _p(frames, _u8p), _p(Kd, _f64p), _p(gt, _f64p))
    if not hnd:
        raise RuntimeError("orc_pipeline_run: a plugin error ended the run")
    res = PipelineResult(lib, "orc_pipeline_", hnd, want_features)
    lib.orc_pipeline_free(hnd)
    return res


def _ext(o):
    lib = o.lib
    return lib


def ba_residuals(cams, pts, obs, cam_idx, pt_idx, K):
    lib = load().lib
    cams = np.ascontiguousarray(cams, np.float64).reshape(-1, 6)
    pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
    ci = np.ascontiguousarray(cam_idx, np.int32)
    pi = np.ascontiguousarray(pt_idx, np.int32)
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    n = obs.shape[0]
    r = np.zeros((n, 2))
    J = np.zeros((n, 2, 9))
    lib.orc_ba_residuals(_p(cams, _f64p), cams.shape[0], _p(pts, _f64p), pts.shape[0], _p(obs, _f64p), _p(ci, _i32p),
                         _p(pi, _i32p), n, _p(Kd, _f64p), _p(r, _f64p), _p(J, _f64p))
    return r, J


def ba_solve(cams, pts, obs, cam_idx, pt_idx, K, huber=1.0, max_iterations=5):
    lib = load().lib
    cams = np.array(cams, np.float64).reshape(-1, 6).copy()
    pts = np.array(pts, np.float64).reshape(-1, 3).copy()
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 2)
    ci = np.ascontiguousarray(cam_idx, np.int32)
    pi = np.ascontiguousarray(pt_idx, np.int32)
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    s = np.zeros(5)
    lib.orc_ba_solve(_p(cams, _f64p), cams.shape[0], _p(pts, _f64p), pts.shape[0], _p(obs, _f64p), _p(ci, _i32p),
                     _p(pi, _i32p), obs.shape[0], _p(Kd, _f64p), C.c_double(huber), max_iterations, _p(s, _f64p))
    return cams, pts, dict(initial_cost=s[0], final_cost=s[1], iterations=int(s[2]), successful_steps=int(s[3]),
                           termination=int(s[4]))


def triangulate_candidates(q1, q2, P1x4, mask_in):
    lib = load().lib
    q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 2)
    n = q1.shape[0]
    P = np.ascontiguousarray(P1x4, np.float64).reshape(48)
    mi = np.ascontiguousarray(mask_in, np.uint8).reshape(n)
    Q = np.zeros((4, 4, n), np.float64)
    mask = np.zeros((4, n), np.uint8)
    good = np.zeros(4, np.int32)
    lib.orc_triangulate_candidates(_p(q1, _f64p), _p(q2, _f64p), n, _p(P, _f64p), _p(mi, _u8p), _p(Q, _f64p), _p(mask, _u8p),
                                   _p(good, _i32p))
    return Q, mask, good


def pnp_ransac(obj, img, K, rvec, tvec, iterations=100, reproj_err=8.0, confidence=0.99):
    lib = load().lib
    o = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
    i2 = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    m = o.shape[0]
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    rv = np.array(rvec, np.float64).reshape(3).copy()
    tv = np.array(tvec, np.float64).reshape(3).copy()
    inl = np.zeros(max(m, 1), np.int32)
    hu = C.c_int()
    n = lib.orc_pnp_ransac(_p(o, _f32p), _p(i2, _f32p), m, _p(Kd, _f64p), _p(rv, _f64p), _p(tv, _f64p), iterations,
                           C.c_float(reproj_err), C.c_double(confidence), _p(inl, _i32p), C.byref(hu))
    return rv, tv, inl[:max(n, 0)].copy(), hu.value


def pnp_hypotheses(obj, img, K, iterations=100, reproj_err=8.0):
    """models (iterations, 6: rvec, tvec) and inlier counts of every RANSAC hypothesis (no adaptive cut-off)"""
    lib = load().lib
    o = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
    i2 = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    models = np.zeros((iterations, 6), np.float64)
    counts = np.zeros(iterations, np.int32)
    lib.orc_pnp_hypotheses(_p(o, _f32p), _p(i2, _f32p), o.shape[0], _p(Kd, _f64p), iterations, C.c_float(reproj_err),
                           _p(models, _f64p), _p(counts, _i32p))
    return models, counts


def rodrigues_v2m(r):
    R = np.zeros(9)
    load().lib.orc_rodrigues_v2m(_p(np.ascontiguousarray(r, np.float64), _f64p), _p(R, _f64p))
    return R.reshape(3, 3)


def rodrigues_m2v(R):
    r = np.zeros(3)
    load().lib.orc_rodrigues_m2v(_p(np.ascontiguousarray(R, np.float64).reshape(9), _f64p), _p(r, _f64p))
    return r
