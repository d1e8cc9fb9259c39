"""The reference's alternative front-end plugins in the oracle: kNNFeatureMatcher (arithmetic in the reference's
own source -> pinned by twins written from kNNFeatureMatcher.cpp:63-122) and cv::FAST (published algorithm -> checked against the
DEFINITION of the segment test and of the corner score, not against the optimised formulas the restatement uses)."""
import math

import numpy as np
import pytest


# ---- kNNFeatureMatcher twins (from the reference source) ----------------------------------------------------------------------
def compare_features(src, cmp, sx, sy, cx, cy, window=15):                     # kNNFeatureMatcher.cpp:103-122
    h, w = src.shape
    _win = math.ceil(np.float32(window) / np.float32(2.0))
    err = np.float32(0)
    for x in range(-_win, _win + 1):
        for y in range(-_win, _win + 1):
            if sx + x < 0 or sy + y < 0 or cx + x < 0 or cy + y < 0 or sx + x >= w or sy + y >= h or cx + x >= w or cy + y >= h:
                continue
            d = np.float32(src[sy + y, sx + x]) - np.float32(cmp[cy + y, cx + x])
            err = np.float32(float(err) + float(d) ** 2)                          # err += pow(float, 2): double addition, float store
    return np.float32(math.sqrt(float(err)) / float(window) ** 2)


def nearest_neighbours(f, feats, n=7):                                          # :63-101; feats: list of (col, row)
    vec, nearest = [], (0, 0)                                                   # default Feature: column 0, row 0
    for _ in range(n):
        dist = 0.0
        for ff in feats:
            if f != ff and all(ff != fff for fff in vec):
                _d = float(max(abs(f[0] - ff[0]), abs(f[1] - ff[1])))           # Feature::distance (Chebyshev)
                if _d < dist or dist == 0:
                    dist, nearest = _d, ff
        vec.append(nearest)
    return vec


def knn_twin(src, cmp, src_xy, cmp_xy, n=7, window=15):
    best, errs = [], []
    cands = [tuple(int(v) for v in c) for c in cmp_xy]
    for f in src_xy:
        f = (int(f[0]), int(f[1]))
        err, pick = np.float32(0), None
        for ff in nearest_neighbours(f, cands, n):
            e = compare_features(src, cmp, f[0], f[1], ff[0], ff[1], window)
            if e < err or err == 0:
                err, pick = e, ff
        best.append(pick)
        errs.append(err)
    return best, np.array(errs, np.float32)


@pytest.mark.parametrize("seed,m", [(1, 60), (2, 5), (3, 0), (4, 9), (5, 200)])
def test_knn_matches_twin_written_from_the_reference(orc, seed, m):
    rng = np.random.default_rng(seed)
    h, w = 60, 90
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    b = np.roll(a, (1, 2), (0, 1)) if seed % 2 else a.copy()           # identical images: err == 0 (the "unset" quirk of :25)
    src_xy = np.stack([rng.integers(0, w, 40), rng.integers(0, h, 40)], 1)
    cmp_xy = np.unique(np.stack([rng.integers(0, w, m), rng.integers(0, h, m)], 1), axis=0) if m else np.zeros((0, 2), int)
    rng.shuffle(cmp_xy)
    if m >= 9:
        cmp_xy[0] = src_xy[0]                                           # a candidate at the source's own coordinates is skipped (f != ff)
        src_xy[1] = (0, 0); src_xy[2] = (w - 1, h - 1)                 # windows cut by the image border
    got_best, got_err = orc.knn_match(a, b, src_xy, cmp_xy)
    want_best, want_err = knn_twin(a, b, src_xy, cmp_xy)
    for i, (g, wb) in enumerate(zip(got_best, want_best)):
        gxy = (0, 0) if g < 0 else tuple(int(v) for v in cmp_xy[g])
        assert gxy == wb, f"feature {i}: best fit {gxy} vs {wb}"
        if g < 0:
            assert wb == (0, 0)
    assert np.array_equal(got_err, want_err)


def test_knn_window_error_float_semantics(orc):
    """the accumulator is a float fed through double additions: sums above 2^24 round per term, in x-outer / y-inner order"""
    import ctypes as C
    a = np.zeros((40, 40), np.uint8)
    b = np.full((40, 40), 255, np.uint8)
    b[::3, ::2] = 254
    orc.lib.orc_knn_compare.restype = C.c_float
    got = orc.lib.orc_knn_compare(a.ctypes.data_as(C.POINTER(C.c_uint8)), b.ctypes.data_as(C.POINTER(C.c_uint8)), 40, 40, 20, 20, 20, 20, 15)
    assert np.float32(got) == compare_features(a, b, 20, 20, 20, 20)
    assert got > 2 ** 24 ** 0.5 / 225 * 0.9


# ---- FAST: the definition ------------------------------------------------------------------------------------------------------
CIRCLE = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def is_corner(img, x, y, t):
    v = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in CIRCLE]
    for sign in (1, -1):
        flags = [(v - p) * sign > t for p in ring]          # darker than v - t / brighter than v + t
        run = 0
        for k in range(16 + 8):
            run = run + 1 if flags[k % 16] else 0
            if run >= 9:
                return True
    return False


def fast_definition(img, t, nonmax=True):
    h, w = img.shape
    score = np.zeros((h, w), int)
    corner = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(img, x, y, t):
                corner[y, x] = True
                s = t
                while s + 1 <= 255 and is_corner(img, x, y, s + 1):
                    s += 1
                score[y, x] = s                               # the largest threshold for which the pixel is still a corner
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if corner[y, x]:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if not nonmax or score[y, x] > nb.max():
                    out.append((x, y, float(score[y, x]) if nonmax else 0.0))
    return out


@pytest.mark.parametrize("seed,shape,t", [(1, (40, 52), 10), (2, (33, 35), 25), (3, (48, 40), 0), (4, (30, 60), 60)])
def test_fast_matches_the_definition(orc, seed, shape, t):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    img[10:20, 12:25] = 200                                    # some structure: a bright block on noise
    img[22:30, 5:15] = 30
    for nonmax in (True, False):
        want = fast_definition(img, t, nonmax)
        xy, rs = orc.fast9_cell(img, (0, 0, shape[1], shape[0]), 100000, threshold=t, nonmax=nonmax)
        got = [(int(x), int(y), float(r)) for (x, y), r in zip(xy, rs)]
        assert got == want
        k = max(1, len(want) // 3)                            # the adapter keeps the first `max` in raster order
        xy, rs = orc.fast9_cell(img, (0, 0, shape[1], shape[0]), k, threshold=t, nonmax=nonmax)
        assert [(int(x), int(y)) for x, y in xy] == [(x, y) for x, y, _ in want[:k]]
    # a sub-view is an image of its own (cv::FAST never reads outside its Mat): same as running on the cropped copy
    cell = (5, 4, shape[1] - 11, shape[0] - 9)
    crop = np.ascontiguousarray(img[cell[1]:cell[1] + cell[3], cell[0]:cell[0] + cell[2]])
    a = orc.fast9_cell(img, cell, 1000, threshold=t)
    b = orc.fast9_cell(crop, (0, 0, cell[2], cell[3]), 1000, threshold=t)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert len(orc.fast9_cell(img, cell, 0, threshold=t)[0]) == 0


def test_cpu_pipeline_with_fast_and_knn_plugins(pmv):
    """the alternative plugin pairs run end to end in the oracle pipeline: FAST + LK, and kNN over FAST"""
    import orc_binding as ob
    w, h, f = 620, 188, 355.0
    frames, gt = pmv.synth_sequence(1004, 0, 16, w, h, f, f, 310.0, 94.0, nthreads=8)
    K = np.array([f, 0, 310.0, 0, f, 94.0, 0, 0, 1.0])
    r = ob.run_pipeline(frames, K, gt, min_tracked=200, tol=75, bundle_size=3, n_threads=4, extractor=2)
    assert r.stats["lk_calls"] == 15 - r.stats["init_offset"] and r.poses.shape[0] >= 8 and all(len(a) > 0 for a in r.features[:5])
    r2 = ob.run_pipeline(frames, K, gt, min_tracked=200, tol=75, bundle_size=3, n_threads=4, extractor=2, matcher=1)
    assert r2.poses.shape[0] >= 8 and len(r2.features) == len(r.features)
    r3 = ob.run_pipeline(frames, K, gt, min_tracked=200, tol=75, bundle_size=3, n_threads=4, extractor=2, matcher=1, threaded=1)
    assert np.array_equal(r2.poses, r3.poses)
