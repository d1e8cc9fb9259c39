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

    def gftt_cell(self, img, cell, max_corners, quality=0.01, min_dist=5.0, want_eig=False):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        x0, y0, cw, ch = [int(v) for v in cell]
        xy = np.zeros((max(max_corners, 1), 2), np.int32)
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
        if not os.path.exists(so) or os.environ.get("PMV_REBUILD_ORACLE"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        _cached = Oracle(C.CDLL(so))
    return _cached
