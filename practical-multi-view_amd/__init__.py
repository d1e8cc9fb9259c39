"""practical-multi-view_amd — MI355X (gfx950) visual-odometry hot path behind the reference's plugin roles.

Python here is plumbing only: a ctypes binding of the C ABI in include/pmv_hip.h (the product is the HIP library
`libpmv_hip.so` built by build.py) plus thin mirrors of the reference's plugin roles used by tests and bench.py.
There is NO CPU fallback: creating a Context without a gfx950 device raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


class PmvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pmv error {code}: {msg}")
        self.code = code


class BaSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int),
                ("successful_steps", C.c_int), ("termination", C.c_int)]


# every symbol include/pmv_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "pmv_ctx_create", "pmv_ctx_destroy", "pmv_last_error", "pmv_sync",
    "pmv_frame_upload", "pmv_frame_upload_bgr", "pmv_frames_stage", "pmv_frames_build", "pmv_frames_stream_begin", "pmv_frames_stream_end", "pmv_frame_get_level", "pmv_frame_get_level_padded", "pmv_frame_num_levels",
    "pmv_detect_gftt", "pmv_detect_shitomasi", "pmv_detect_fast", "pmv_knn_match", "pmv_debug_gftt_response", "pmv_debug_shitomasi_response",
    "pmv_lk_track", "pmv_set_ba_mode", "pmv_pnp_ransac", "pmv_debug_pnp_hypotheses", "pmv_debug_ba_stamps", "pmv_debug_lk_stamps", "pmv_ba_residuals", "pmv_ba_solve", "pmv_triangulate_candidates", "pmv_triangulate_candidates_ahead", "pmv_fivepoint_hypotheses",
    "pmv_record_enable", "pmv_record_count", "pmv_record_size", "pmv_record_get",
    "pmv_prof_enable", "pmv_prof_select", "pmv_prof_kernel_count", "pmv_lk_counters", "pmv_prof_kernel_name", "pmv_prof_read",
    "pmv_pipeline_run", "pmv_pipeline_run_streamed", "pmv_pipeline_run_batch", "pmv_batch_stats", "pmv_pipeline_free", "pmv_pipeline_release", "pmv_pipeline_drain", "pmv_pipeline_num_poses", "pmv_pipeline_get_poses", "pmv_pipeline_num_frames",
    "pmv_pipeline_frame_feature_count", "pmv_pipeline_get_frame_features", "pmv_pipeline_stats_count", "pmv_pipeline_get_stats",
]


GFTT_UNLIMITED_CAP = 4096   # PMV_GFTT_UNLIMITED_CAP of include/pmv_hip.h


class PipelineParams(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("w", C.c_int), ("h", C.c_int), ("min_tracked_features", C.c_int),
                ("tracked_features_tol", C.c_int), ("init_frames", C.c_int), ("bundle_size", C.c_int),
                ("ba_iterations", C.c_int), ("extractor", C.c_int), ("threaded", C.c_int), ("n_threads", C.c_int),
                ("build_pyramids", C.c_int), ("matcher", C.c_int), ("device_fivepoint", C.c_int)]


STAT_KEYS = ["lk_calls", "lk_points", "detect_calls", "pnp_calls", "pnp_points", "tri_calls", "ba_calls", "ba_obs",
             "ba_points", "heuristic_motion", "seconds", "init_offset", "n_landmarks", "scale", "t_lk", "t_detect", "t_pnp", "t_tri",
             "t_ba", "t_pnp_kernel", "t_ba_kernel", "t_tri_essential", "t_tri_pose", "tri_hypotheses", "tri_ahead"]


class PipelineResult:
    """poses (n,12: R row-major then t), per-frame (column,row,landmark) triples in container order, run statistics"""

    def __init__(self, lib, handle, want_features=True):
        n = lib.pmv_pipeline_num_poses(handle)
        self.poses = np.zeros((n, 12), np.float64)
        if n:
            lib.pmv_pipeline_get_poses(handle, _p(self.poses, _f64p))
        self.features = []
        if want_features:
            for k in range(lib.pmv_pipeline_num_frames(handle)):
                c = lib.pmv_pipeline_frame_feature_count(handle, k)
                a = np.zeros((c, 3), np.int32)
                if c:
                    lib.pmv_pipeline_get_frame_features(handle, k, _p(a, _i32p))
                self.features.append(a)
        st = np.zeros(lib.pmv_pipeline_stats_count(), np.float64)   # the library says how many doubles it writes
        assert len(st) >= len(STAT_KEYS)
        lib.pmv_pipeline_get_stats(handle, _p(st, _f64p))
        self.stats = dict(zip(STAT_KEYS, [float(v) for v in st[:len(STAT_KEYS)]]))
        self._deferred = None   # (lib, handle) when the caller asked to free the native result later (defer_free)

    def free(self):
        """frees the native result of a pipeline_run(..., defer_free=True) call (idempotent)"""
        if self._deferred:
            lib, handle = self._deferred
            self._deferred = None
            lib.pmv_pipeline_free(handle)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def parse_record(buf):
    """one blob of the call log (layout: include/pmv_hip.h, "call log")"""
    pos = [0]
    raw = buf.tobytes()

    def take(dtype, count):
        nb = np.dtype(dtype).itemsize * count
        a = np.frombuffer(raw[pos[0]:pos[0] + nb], dtype=dtype).copy()
        pos[0] += nb
        return a
    kind = int(take(np.int32, 1)[0])
    if kind == 0:
        m, iters, nin = [int(v) for v in take(np.int32, 3)]
        r = dict(kind="pnp", m=m, iterations=iters, obj=take(np.float32, 3 * m).reshape(m, 3), img=take(np.float32, 2 * m).reshape(m, 2),
                 K=take(np.float64, 9), rvec_in=take(np.float64, 3), tvec_in=take(np.float64, 3))
        opt = take(np.float64, 2)
        r.update(reproj_err=float(opt[0]), confidence=float(opt[1]), rvec=take(np.float64, 3), tvec=take(np.float64, 3), inliers=take(np.int32, nin))
    elif kind == 1:
        nc, npnt, nobs, iters = [int(v) for v in take(np.int32, 4)]
        r = dict(kind="ba", nc=nc, np=npnt, n_obs=nobs, max_iterations=iters, cams_in=take(np.float64, 6 * nc).reshape(nc, 6),
                 pts_in=take(np.float64, 3 * npnt).reshape(npnt, 3), obs=take(np.float64, 2 * nobs).reshape(nobs, 2), K=take(np.float64, 9),
                 huber=float(take(np.float64, 1)[0]), cam_idx=take(np.int32, nobs), pt_idx=take(np.int32, nobs))
        r.update(cams=take(np.float64, 6 * nc).reshape(nc, 6), pts=take(np.float64, 3 * npnt).reshape(npnt, 3))
        sm = take(np.float64, 5)
        r.update(initial_cost=float(sm[0]), final_cost=float(sm[1]), iterations=int(sm[2]), successful_steps=int(sm[3]), termination=int(sm[4]))
    elif kind == 2:
        n = int(take(np.int32, 1)[0])
        r = dict(kind="dlt", n=n, q1=take(np.float64, 2 * n).reshape(n, 2), q2=take(np.float64, 2 * n).reshape(n, 2), P1x4=take(np.float64, 48),
                 mask_in=take(np.uint8, n), Q=take(np.float64, 16 * n).reshape(4, 4, n), mask=take(np.uint8, 4 * n).reshape(4, n), good=take(np.int32, 4))
    else:
        raise ValueError(f"unknown record kind {kind}")
    assert pos[0] == len(buf), "record length does not match its header"
    return r


_lib = None
_synth = None


def lib_path():
    return os.path.join(HERE, "libpmv_hip.so")


def load_library():
    """dlopen the product library (must have been built: __graft_entry__.build() or build.build_hip())."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise PmvError(-1, f"{p} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        _lib.pmv_last_error.restype = C.c_char_p
        _lib.pmv_last_error.argtypes = [C.c_void_p]
        if hasattr(_lib, "pmv_pipeline_run"):
            _lib.pmv_pipeline_run.argtypes = [C.c_void_p, C.POINTER(PipelineParams), _f64p, _f64p, C.POINTER(C.c_void_p)]
            _lib.pmv_pipeline_run_streamed.argtypes = [C.c_void_p, C.POINTER(PipelineParams), _f64p, _f64p, _u8p, C.POINTER(C.c_void_p)]
            _lib.pmv_pipeline_free.argtypes = [C.c_void_p]
            _lib.pmv_pipeline_release.argtypes = [C.c_void_p]
            _lib.pmv_pipeline_release.restype = None
            _lib.pmv_pipeline_drain.argtypes = []
            _lib.pmv_pipeline_drain.restype = None
            _lib.pmv_pipeline_num_poses.argtypes = [C.c_void_p]
            _lib.pmv_pipeline_get_poses.argtypes = [C.c_void_p, _f64p]
            _lib.pmv_pipeline_num_frames.argtypes = [C.c_void_p]
            _lib.pmv_pipeline_frame_feature_count.argtypes = [C.c_void_p, C.c_int]
            _lib.pmv_pipeline_get_frame_features.argtypes = [C.c_void_p, C.c_int, _i32p]
            _lib.pmv_pipeline_get_stats.argtypes = [C.c_void_p, _f64p]
    return _lib


def load_synth():
    global _synth
    if _synth is None:
        _synth = C.CDLL(_build.build_synth())
    return _synth


def _p(a, t):
    return a.ctypes.data_as(t)


def synth_sequence(seed, first, n, w, h, fx, fy, cx, cy, nthreads=8):
    """n deterministic synthetic KITTI-like frames (n, h, w) uint8 + their KITTI pose rows (n, 12)."""
    s = load_synth()
    out = np.empty((n, h, w), np.uint8)
    s.pmv_synth_sequence(C.c_uint64(seed), first, n, w, h, C.c_double(fx), C.c_double(fy), C.c_double(cx),
                         C.c_double(cy), _p(out, _u8p), nthreads)
    poses = np.empty((n, 12), np.float64)
    for i in range(n):
        s.pmv_synth_pose(C.c_uint64(seed), first + i, _p(poses[i], _f64p))
    return out, poses


def grid_cells(w, h, gw=255, gh=255):
    """OdometryPipeline::getGridROI (reference OdometryPipeline.cpp:674-693): row-major cells (x0, y0, cw, ch)."""
    cells = []
    for r in range(0, h, gh):
        for c in range(0, w, gw):
            cells.append((c, r, min(gw, w - c), min(gh, h - r)))
    return np.asarray(cells, np.int32)


class Context:
    """Opaque device context (owns HBM frame slots, workspaces and the front-end/back-end HIP streams)."""

    def __init__(self, max_w, max_h, n_slots=2, max_tracks=4096, max_ba_cams=32, max_ba_points=8192,
                 max_ba_obs=65536, device=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.pmv_ctx_create(C.byref(self.h), device, max_w, max_h, n_slots, max_tracks, max_ba_cams,
                                     max_ba_points, max_ba_obs)
        if rc != 0:
            raise PmvError(rc, self.lib.pmv_last_error(None).decode())

    def close(self):
        if self.h:
            self.lib.pmv_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise PmvError(rc, self.lib.pmv_last_error(self.h).decode())

    def sync(self):
        self._ck(self.lib.pmv_sync(self.h))

    # ---- frames ----
    def frame_upload(self, slot, gray):
        g = np.ascontiguousarray(gray, np.uint8)
        self._ck(self.lib.pmv_frame_upload(self.h, slot, _p(g, _u8p), g.shape[1], g.shape[0], g.shape[1]))

    def frame_upload_bgr(self, slot, bgr):
        """(h, w, 3) uint8 BGR image as cv::imread(IMREAD_COLOR) gives it: BGR2GRAY on the device, then the pyramid"""
        b = np.ascontiguousarray(bgr, np.uint8)
        assert b.ndim == 3 and b.shape[2] == 3
        self._ck(self.lib.pmv_frame_upload_bgr(self.h, slot, _p(b, _u8p), b.shape[1], b.shape[0], 3 * b.shape[1]))

    def frames_stage(self, first_slot, frames):
        f = np.ascontiguousarray(frames, np.uint8)
        self._ck(self.lib.pmv_frames_stage(self.h, first_slot, f.shape[0], _p(f, _u8p), f.shape[2], f.shape[1]))

    def frames_stream_begin(self, first_slot, frames):
        """start streaming host frames (n, h, w) uint8 into slots first_slot..; `frames` must stay alive until frames_stream_end()"""
        f = np.ascontiguousarray(frames, np.uint8)
        self._stream_src = f
        self._ck(self.lib.pmv_frames_stream_begin(self.h, first_slot, f.shape[0], _p(f, _u8p), f.shape[2], f.shape[1]))

    def frames_stream_end(self):
        self._ck(self.lib.pmv_frames_stream_end(self.h))
        self._stream_src = None

    def frames_build(self, first_slot, n):
        self._ck(self.lib.pmv_frames_build(self.h, first_slot, n))

    def num_levels(self, slot):
        return self.lib.pmv_frame_num_levels(self.h, slot)

    def get_level(self, slot, level, max_w, max_h):
        out = np.zeros(max_w * max_h, np.uint8)
        w, h = C.c_int(), C.c_int()
        self._ck(self.lib.pmv_frame_get_level(self.h, slot, level, _p(out, _u8p), C.byref(w), C.byref(h)))
        return out[: w.value * h.value].reshape(h.value, w.value).copy()

    def get_level_padded(self, slot, level, max_w, max_h):
        """the level with its 64-pixel BORDER_REFLECT_101 frame"""
        out = np.zeros((max_w + 128) * (max_h + 128), np.uint8)
        w, h = C.c_int(), C.c_int()
        self._ck(self.lib.pmv_frame_get_level_padded(self.h, slot, level, _p(out, _u8p), C.byref(w), C.byref(h)))
        return out[: w.value * h.value].reshape(h.value, w.value).copy()

    # ---- BaseFeatureExtractor role ----
    def detect_gftt(self, slot, cells, max_per_cell, quality=0.01, min_dist=5.0):
        cells = np.ascontiguousarray(cells, np.int32).reshape(-1, 4)
        n = cells.shape[0]
        cap = max_per_cell if max_per_cell > 0 else GFTT_UNLIMITED_CAP   # max_per_cell <= 0: no limit (cv::goodFeaturesToTrack)
        xy = np.zeros((n, cap, 2), np.int32)
        cnt = np.zeros(n, np.int32)
        self._ck(self.lib.pmv_detect_gftt(self.h, slot, _p(cells, _i32p), n, max_per_cell, C.c_double(quality),
                                          C.c_double(min_dist), _p(xy, _i32p), _p(cnt, _i32p)))
        return [xy[i, : cnt[i]].copy() for i in range(n)]

    def detect_shitomasi(self, slot, cells, max_per_cell, quality=0.4):
        cells = np.ascontiguousarray(cells, np.int32).reshape(-1, 4)
        n = cells.shape[0]
        xy = np.zeros((n, max(max_per_cell, 1), 2), np.int32)
        sc = np.zeros((n, max(max_per_cell, 1)), np.float64)
        cnt = np.zeros(n, np.int32)
        self._ck(self.lib.pmv_detect_shitomasi(self.h, slot, _p(cells, _i32p), n, max_per_cell, C.c_double(quality),
                                               _p(xy, _i32p), _p(sc, _f64p), _p(cnt, _i32p)))
        return [(xy[i, : cnt[i]].copy(), sc[i, : cnt[i]].copy()) for i in range(n)]

    def detect_fast(self, slot, cells, max_per_cell, threshold=10, nonmax=True):
        """cv::FAST per view (views may be as large as the frame): [(xy (k,2) int32, response (k,) float32)] per cell"""
        cells = np.ascontiguousarray(cells, np.int32).reshape(-1, 4)
        n = cells.shape[0]
        cap = max(max_per_cell, 1)
        xy = np.zeros((n, cap, 2), np.int32)
        rs = np.zeros((n, cap), np.float32)
        cnt = np.zeros(n, np.int32)
        self._ck(self.lib.pmv_detect_fast(self.h, slot, _p(cells, _i32p), n, max_per_cell, threshold, 1 if nonmax else 0, _p(xy, _i32p), _p(rs, _f32p),
                                          _p(cnt, _i32p)))
        return [(xy[i, : cnt[i]].copy(), rs[i, : cnt[i]].copy()) for i in range(n)]

    def knn_match(self, src_slot, cmp_slot, src_xy, cmp_xy, neighbours=7, window=15):
        """kNNFeatureMatcher's arithmetic: (best candidate index or -1, window error) per source feature"""
        s = np.ascontiguousarray(src_xy, np.int32).reshape(-1, 2)
        c = np.ascontiguousarray(cmp_xy, np.int32).reshape(-1, 2)
        best = np.zeros(max(len(s), 1), np.int32)
        err = np.zeros(max(len(s), 1), np.float32)
        self._ck(self.lib.pmv_knn_match(self.h, src_slot, cmp_slot, _p(s, _i32p), len(s), _p(c, _i32p), len(c), neighbours, window, _p(best, _i32p), _p(err, _f32p)))
        return best[: len(s)].copy(), err[: len(s)].copy()

    def gftt_response(self, slot, cell):
        cell = np.ascontiguousarray(cell, np.int32)
        out = np.zeros((cell[3], cell[2]), np.float32)
        self._ck(self.lib.pmv_debug_gftt_response(self.h, slot, _p(cell, _i32p), _p(out, _f32p)))
        return out

    def shitomasi_response(self, slot, cell):
        cell = np.ascontiguousarray(cell, np.int32)
        out = np.zeros((cell[3], cell[2]), np.float64)
        self._ck(self.lib.pmv_debug_shitomasi_response(self.h, slot, _p(cell, _i32p), _p(out, _f64p)))
        return out

    # ---- BaseFeatureMatcher role ----
    def lk_track(self, prev_slot, next_slot, prev_xy):
        p = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
        n = p.shape[0]
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self._ck(self.lib.pmv_lk_track(self.h, prev_slot, next_slot, _p(p, _f32p), n, _p(out, _f32p), _p(st, _u8p),
                                       _p(err, _f32p)))
        return out, st, err

    # ---- BasePnPSolver role ----
    def pnp_ransac(self, obj_xyz, img_xy, K, rvec, tvec, iterations=100, reproj_err=8.0, confidence=0.99):
        o = np.ascontiguousarray(obj_xyz, np.float32).reshape(-1, 3)
        i2 = np.ascontiguousarray(img_xy, np.float32).reshape(-1, 2)
        m = o.shape[0]
        Kd = np.ascontiguousarray(K, np.float64).reshape(9)
        rv = np.array(rvec, np.float64).reshape(3).copy()
        tv = np.array(tvec, np.float64).reshape(3).copy()
        inl = np.zeros(max(m, 1), np.int32)
        nin = C.c_int()
        self._ck(self.lib.pmv_pnp_ransac(self.h, _p(o, _f32p), _p(i2, _f32p), m, _p(Kd, _f64p), _p(rv, _f64p),
                                         _p(tv, _f64p), iterations, C.c_float(reproj_err), C.c_double(confidence),
                                         _p(inl, _i32p), C.byref(nin)))
        return rv, tv, inl[: nin.value].copy()

    def pnp_hypotheses(self, n=100):
        models = np.zeros((n, 6), np.float64)
        counts = np.zeros(n, np.int32)
        self._ck(self.lib.pmv_debug_pnp_hypotheses(self.h, n, _p(models, _f64p), _p(counts, _i32p)))
        return models, counts

    # ---- BaseOptimizer role ----
    def ba_residuals(self, cams, pts, obs_xy, cam_idx, pt_idx, K):
        cams = np.ascontiguousarray(cams, np.float64).reshape(-1, 6)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        obs = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
        ci = np.ascontiguousarray(cam_idx, np.int32)
        pi = np.ascontiguousarray(pt_idx, np.int32)
        Kd = np.ascontiguousarray(K, np.float64).reshape(9)
        n = obs.shape[0]
        r = np.zeros((n, 2), np.float64)
        J = np.zeros((n, 2, 9), np.float64)
        self._ck(self.lib.pmv_ba_residuals(self.h, _p(cams, _f64p), cams.shape[0], _p(pts, _f64p), pts.shape[0],
                                           _p(obs, _f64p), _p(ci, _i32p), _p(pi, _i32p), n, _p(Kd, _f64p),
                                           _p(r, _f64p), _p(J, _f64p)))
        return r, J

    def set_ba_mode(self, mode):
        """0 = multi-kernel LM chain (default), 1 = one workgroup per solve / one launch per batched round"""
        self._ck(self.lib.pmv_set_ba_mode(self.h, int(mode)))

    def ba_solve(self, cams, pts, obs_xy, cam_idx, pt_idx, K, huber=1.0, max_iterations=5):
        cams = np.array(cams, np.float64).reshape(-1, 6).copy()
        pts = np.array(pts, np.float64).reshape(-1, 3).copy()
        obs = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
        ci = np.ascontiguousarray(cam_idx, np.int32)
        pi = np.ascontiguousarray(pt_idx, np.int32)
        Kd = np.ascontiguousarray(K, np.float64).reshape(9)
        s = BaSummary()
        self._ck(self.lib.pmv_ba_solve(self.h, _p(cams, _f64p), cams.shape[0], _p(pts, _f64p), pts.shape[0],
                                       _p(obs, _f64p), _p(ci, _i32p), _p(pi, _i32p), obs.shape[0], _p(Kd, _f64p),
                                       C.c_double(huber), max_iterations, C.byref(s)))
        return cams, pts, s

    def fivepoint_hypotheses(self, q1, q2, samples, thr):
        """one RANSAC round of findEssentialMat on the GPU: models (n_hyp, 10, 9), n_models (n_hyp,), counts (n_hyp, 10)"""
        q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 2)
        q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 2)
        s = np.ascontiguousarray(samples, np.int32).reshape(-1, 5)
        nh = s.shape[0]
        models = np.zeros((nh, 10, 9), np.float64)
        nm = np.zeros(nh, np.int32)
        counts = np.zeros((nh, 10), np.int32)
        self._ck(self.lib.pmv_fivepoint_hypotheses(self.h, _p(q1, _f64p), _p(q2, _f64p), q1.shape[0], _p(s, _i32p), nh, C.c_float(thr), _p(models, _f64p),
                                                   _p(nm, _i32p), _p(counts, _i32p)))
        return models, nm, counts

    def triangulate_candidates(self, q1, q2, P1x4, mask_in):
        """DLT triangulation + cheirality of cv::recoverPose's four candidates: returns Q (4,4,n), mask (4,n), good (4,)"""
        q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 2)
        q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 2)
        n = q1.shape[0]
        P = np.ascontiguousarray(P1x4, np.float64).reshape(48)
        mi = np.ascontiguousarray(mask_in, np.uint8).reshape(n)
        Q = np.zeros((4, 4, n), np.float64)
        mask = np.zeros((4, n), np.uint8)
        good = np.zeros(4, np.int32)
        self._ck(self.lib.pmv_triangulate_candidates(self.h, _p(q1, _f64p), _p(q2, _f64p), n, _p(P, _f64p), _p(mi, _u8p),
                                                     _p(Q, _f64p), _p(mask, _u8p), _p(good, _i32p)))
        return Q, mask, good

    # ---- call log (teacher-forced replay) ----
    def record_enable(self, on=True):
        self._ck(self.lib.pmv_record_enable(self.h, 1 if on else 0))

    def records(self):
        """the logged back-end calls as dicts (kind 'pnp' | 'ba' | 'dlt') with numpy views of inputs and outputs"""
        self.lib.pmv_record_size.restype = C.c_longlong
        self.lib.pmv_record_size.argtypes = [C.c_void_p, C.c_int]
        self.lib.pmv_record_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        out = []
        for i in range(self.lib.pmv_record_count(self.h)):
            nb = self.lib.pmv_record_size(self.h, i)
            buf = np.zeros(nb, np.uint8)
            self._ck(self.lib.pmv_record_get(self.h, i, buf.ctypes.data_as(C.c_void_p), nb))
            out.append(parse_record(buf))
        return out

    # ---- per-kernel HIP-event timing ----
    def prof_enable(self, on=True):
        self._ck(self.lib.pmv_prof_enable(self.h, 1 if on else 0))

    def lk_counters(self, reset=False):
        """(LK iterations, (track, level) passes, tracks) executed by k_lk since the last reset"""
        out = (C.c_ulonglong * 3)()
        self._ck(self.lib.pmv_lk_counters(self.h, out, 1 if reset else 0))
        return int(out[0]), int(out[1]), int(out[2])

    def prof_select(self, names):
        """after prof_enable(True): record only the named kernel classes"""
        self.lib.pmv_prof_kernel_name.restype = C.c_char_p
        mask = 0
        for i in range(self.lib.pmv_prof_kernel_count()):
            if self.lib.pmv_prof_kernel_name(i).decode() in names:
                mask |= 1 << i
        self._ck(self.lib.pmv_prof_select(self.h, C.c_uint(mask)))

    def prof_read(self):
        """{kernel: (launches, total_ms, max_ms)} for every kernel class launched since prof_enable(True)"""
        self.lib.pmv_prof_kernel_name.restype = C.c_char_p
        out = {}
        for i in range(self.lib.pmv_prof_kernel_count()):
            n, tot, mx = C.c_int(), C.c_double(), C.c_double()
            self._ck(self.lib.pmv_prof_read(self.h, i, C.byref(n), C.byref(tot), C.byref(mx)))
            if n.value:
                out[self.lib.pmv_prof_kernel_name(i).decode()] = (n.value, tot.value, mx.value)
        return out

    # ---- whole sequence (OdometryPipeline role) ----
    def pipeline_run(self, n_frames, w, h, K, gt_poses, min_tracked=400, tol=150, init_frames=5, bundle_size=5,
                     ba_iterations=5, extractor=0, threaded=0, build_pyramids=1, want_features=True, n_threads=1, async_free=False,
                     defer_free=False, host_frames=None, matcher=0, device_fivepoint=0):
        """frames 0..n_frames-1 must be staged in slots 0..n_frames-1 (frames_stage) unless host_frames (n, h, w) uint8 is given:
        then they are streamed from host memory while the pipeline runs (pmv_pipeline_run_streamed). n_threads: host threads that
        evaluate the five-point RANSAC hypotheses of the triangulator side by side (the results do not depend on it)"""
        P = PipelineParams(n_frames, w, h, min_tracked, tol, init_frames, bundle_size, ba_iterations, extractor, threaded,
                           n_threads, build_pyramids, matcher, device_fivepoint)
        Kd = np.ascontiguousarray(K, np.float64).reshape(9)
        gt = np.ascontiguousarray(gt_poses, np.float64).reshape(n_frames, 12)
        out = C.c_void_p()
        if host_frames is not None:
            hf = np.ascontiguousarray(host_frames, np.uint8)
            assert hf.shape == (n_frames, h, w)
            self._ck(self.lib.pmv_pipeline_run_streamed(self.h, C.byref(P), _p(Kd, _f64p), _p(gt, _f64p), _p(hf, _u8p), C.byref(out)))
        else:
            self._ck(self.lib.pmv_pipeline_run(self.h, C.byref(P), _p(Kd, _f64p), _p(gt, _f64p), C.byref(out)))
        # Tearing down the ~10^6 host-container nodes of a long run takes ~40 ms and is not part of the path. defer_free: the
        # caller frees later (result.free()); async_free: a background thread does it (pipeline_drain() joins).
        ok = False
        try:
            r = PipelineResult(self.lib, out, want_features)
            ok = True
        finally:
            if not (ok and defer_free):
                if async_free:
                    self.lib.pmv_pipeline_release(out)
                else:
                    self.lib.pmv_pipeline_free(out)
        if defer_free:
            r._deferred = (self.lib, out)
        return r

    def pipeline_run_batch(self, seqs, w, h, K, min_tracked=400, tol=150, init_frames=5, bundle_size=5, ba_iterations=5, extractor=0,
                           build_pyramids=1, want_features=True, defer_free=False, threaded=1, device_fivepoint=0):
        """B independent sequences through batched launches (pmv_pipeline_run_batch). seqs: list of (first_slot, n_frames, gt_poses);
        the frames must be staged in slots first_slot..first_slot+n_frames-1. K: 9 values shared by all, or (B, 9). Returns one
        PipelineResult per sequence (bit-identical to pipeline_run on the same sequence)."""
        B = len(seqs)
        params = (PipelineParams * B)()
        gts = []
        gt_ptrs = (_f64p * B)()
        first = (C.c_int * B)()
        Kd = np.ascontiguousarray(np.broadcast_to(np.asarray(K, np.float64).reshape(-1, 9), (B, 9)))
        for b, (fs, n, gt) in enumerate(seqs):
            params[b] = PipelineParams(n, w, h, min_tracked, tol, init_frames, bundle_size, ba_iterations, extractor, threaded, 1, build_pyramids, 0, device_fivepoint)
            g = np.ascontiguousarray(gt, np.float64).reshape(n, 12)
            gts.append(g)
            gt_ptrs[b] = _p(g, _f64p)
            first[b] = fs
        outs = (C.c_void_p * B)()
        self.lib.pmv_pipeline_run_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(PipelineParams), _f64p, C.POINTER(_f64p), _i32p, C.POINTER(C.c_void_p)]
        self._ck(self.lib.pmv_pipeline_run_batch(self.h, B, params, _p(Kd, _f64p), gt_ptrs, first, outs))
        res = []
        for b in range(B):
            hnd = C.c_void_p(outs[b])
            r = PipelineResult(self.lib, hnd, want_features)
            if defer_free:
                r._deferred = (self.lib, hnd)
            else:
                self.lib.pmv_pipeline_free(hnd)
            res.append(r)
        return res

    def batch_stats(self):
        """per combiner of the batch engine (lk, det, pnp, ba, dlt): launch rounds, requests served, CPU seconds of the thread, wall seconds processing batches / of that waiting for the GPU"""
        cnt = (C.c_longlong * 10)()
        t = (C.c_double * 15)()
        self.lib.pmv_batch_stats(self.h, cnt, t)
        out = {}
        for r, name in enumerate(("lk", "det", "pnp", "ba", "dlt")):
            out[name] = dict(launches=int(cnt[2 * r]), requests=int(cnt[2 * r + 1]), cpu_s=t[3 * r], work_s=t[3 * r + 1], sync_s=t[3 * r + 2])
        return out

    def pipeline_drain(self):
        self.lib.pmv_pipeline_drain()
