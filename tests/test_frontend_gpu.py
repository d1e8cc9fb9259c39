"""GPU parity (through the C ABI) of the front-end kernels against the oracle: pyramid bytes, GFTT / ShiTomasi corner
lists and response maps, LK positions/status/err.  Bar: bit-exact (integer / index work and fixed-order float32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KITTI07 = dict(w=1226, h=370, fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104)
KITTI00 = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)


def _frames(pmv, cfg, n, seed=1007, first=0):
    f, _ = pmv.synth_sequence(seed, first, n, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"])
    return f


@pytest.mark.parametrize("cfg", [KITTI07, KITTI00, dict(w=321, h=163, fx=200., fy=200., cx=160., cy=80.)])
def test_pyramid_matches_oracle(pmv, orc, gpu_ctx_factory, cfg):
    fr = _frames(pmv, cfg, 1)[0]
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=1)
    ctx.frame_upload(0, fr)
    nl = ctx.num_levels(0)
    ref = fr
    for l in range(nl + 1):
        got = ctx.get_level(0, l, cfg["w"], cfg["h"])
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), f"level {l} differs"
        ref = orc.pyr_down(ref)
    # level-count rule of buildOpticalFlowPyramid: next level would be <= 32 in a dimension
    assert min(ref.shape) <= 32 or nl == 4


def test_colour_frame_upload_matches_oracle(pmv, orc, gpu_ctx_factory):
    """Frame::Frame(file) / Frame::init (Frame.cpp:33,40-41): a BGR image through pmv_frame_upload_bgr gives the pyramid of the oracle's
    BGR2GRAY image, at an odd width (unaligned rows, partial last thread) too; a gray image sent as BGR equals pmv_frame_upload."""
    rng = np.random.default_rng(9)
    for w, h in ((1241, 376), (333, 97)):
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ctx = gpu_ctx_factory(w, h, n_slots=2, max_tracks=64)
        ctx.frame_upload_bgr(0, bgr)
        gray = orc.bgr2gray(bgr)
        ctx.frame_upload(1, gray)
        for l in range(ctx.num_levels(0) + 1):
            assert np.array_equal(ctx.get_level(0, l, w, h), ctx.get_level(1, l, w, h)), (w, h, l)
            assert np.array_equal(ctx.get_level_padded(0, l, w, h), ctx.get_level_padded(1, l, w, h))
        assert np.array_equal(ctx.get_level(0, 0, w, h), gray)
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ctx.frame_upload_bgr(0, np.repeat(g[..., None], 3, axis=2))
        assert np.array_equal(ctx.get_level(0, 0, w, h), g)


@pytest.mark.parametrize("w,h", [(1241, 376), (1226, 370), (321, 163), (224, 131), (113, 97), (111, 80), (100, 66), (97, 67)])
def test_pyramid_reflect101_frame(pmv, orc, gpu_ctx_factory, w, h):
    """every level carries a 64-pixel BORDER_REFLECT_101 frame (cv::buildOpticalFlowPyramid pads each level; LK reads it for windows
    that leave the image): the frame bytes equal numpy's reflect padding of the oracle's level, for widths on both sides of the kernels'
    wide / narrow border paths (112) and for rows whose right edge falls on every position inside a 16-byte group"""
    rng = np.random.default_rng(w * 1000 + h)
    fr = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ctx = gpu_ctx_factory(w, h, n_slots=1)
    ctx.frame_upload(0, fr)
    ref = fr
    for l in range(ctx.num_levels(0) + 1):
        got = ctx.get_level_padded(0, l, w, h)
        want = np.pad(ref, 64, mode="reflect") if min(ref.shape) > 64 else None
        if want is None:   # numpy's reflect needs pad < size; build the REFLECT_101 index map by hand
            def idx(n, size):
                p = np.arange(-64, size + 64)
                while ((p < 0) | (p >= size)).any():
                    p = np.where(p < 0, -p, p); p = np.where(p >= size, 2 * size - 2 - p, p)
                return p
            want = ref[np.ix_(idx(64, ref.shape[0]), idx(64, ref.shape[1]))]
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"level {l}: {np.argwhere(got != want)[:5]}"
        ref = orc.pyr_down(ref)


def test_batched_build_equals_single_upload(pmv, gpu_ctx_factory):
    cfg = KITTI07
    fr = _frames(pmv, cfg, 3)
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=4)
    ctx.frames_stage(0, fr)
    ctx.frames_build(0, 3)
    ctx.frame_upload(3, fr[2])
    for l in range(ctx.num_levels(3) + 1):
        assert np.array_equal(ctx.get_level(2, l, cfg["w"], cfg["h"]), ctx.get_level(3, l, cfg["w"], cfg["h"]))


@pytest.mark.parametrize("cfg,per_cell", [(KITTI07, 40), (KITTI00, 80)])
def test_gftt_matches_oracle(pmv, orc, gpu_ctx_factory, cfg, per_cell):
    fr = _frames(pmv, cfg, 1, seed=1003)[0]
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=1)
    ctx.frame_upload(0, fr)
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    assert len(cells) == 10
    got = ctx.detect_gftt(0, cells, per_cell)
    for c, g in zip(cells, got):
        ref, eig = orc.gftt_cell(fr, c, per_cell, want_eig=True)
        assert np.array_equal(ctx.gftt_response(0, c), eig), "min-eigenvalue map differs"
        assert np.array_equal(g, ref), f"corner list differs in cell {c}"
        assert len(g) > 0


def test_gftt_edge_cases(pmv, orc, gpu_ctx_factory):
    w, h = 300, 280
    rng = np.random.default_rng(5)
    flat = np.full((h, w), 77, np.uint8)                       # no corners at all
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)       # dense candidates, many near-ties
    ties = np.zeros((h, w), np.uint8)
    ties[::16, ::16] = 255                                       # identical responses: tie-break by address
    ctx = gpu_ctx_factory(w, h, n_slots=1)
    cells = pmv.grid_cells(w, h)
    for img in (flat, noise, ties):
        ctx.frame_upload(0, img)
        for mx, md in ((40, 5.0), (500, 5.0), (25, 0.5), (30, 11.3)):
            got = ctx.detect_gftt(0, cells, mx, min_dist=md)
            for c, g in zip(cells, got):
                ref = orc.gftt_cell(img, c, mx, min_dist=md)
                assert np.array_equal(g, ref)
    small = np.asarray([[7, 9, 3, 3], [0, 0, 5, 4]], np.int32)  # tiny cells
    got = ctx.detect_gftt(0, small, 10)
    for c, g in zip(small, got):
        assert np.array_equal(g, orc.gftt_cell(ties, c, 10))


def test_gftt_no_limit_semantics(pmv, orc, gpu_ctx_factory):
    """cv::goodFeaturesToTrack(maxCorners <= 0) returns every corner that survives the min-distance rule; ShiTomasi's loop
    (ShiTomasiFeatureExtractor.cpp:37-44) returns nothing for max <= 0."""
    cfg = KITTI07
    fr = _frames(pmv, cfg, 1, seed=1003)[0]
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=1)
    ctx.frame_upload(0, fr)
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    for mx in (0, -3):
        got = ctx.detect_gftt(0, cells, mx)
        for c, g in zip(cells, got):
            ref = orc.gftt_cell(fr, c, 65536)     # the oracle's "no limit": a bound no cell can reach
            ref0 = orc.gftt_cell(fr, c, mx) if mx == 0 else ref
            assert np.array_equal(ref, ref0)
            assert len(g) > 80 and np.array_equal(g, ref)
        assert all(len(xy) == 0 for xy, _ in ctx.detect_shitomasi(0, cells, mx))
    # more corners than the documented capacity of the no-limit form is an error, not a silent truncation; the next call is clean
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, (cfg["h"], cfg["w"]), dtype=np.uint8)
    ctx.frame_upload(0, noise)
    with pytest.raises(pmv.PmvError) as e:
        ctx.detect_gftt(0, cells, 0, min_dist=0.5)
    assert e.value.code == -6
    got = ctx.detect_gftt(0, cells, 40)
    for c, g in zip(cells, got):
        assert np.array_equal(g, orc.gftt_cell(noise, c, 40))


def test_detector_candidate_lists_larger_than_lds(pmv, orc, gpu_ctx_factory):
    """The reference sorts ALL candidates of a cell; the kernels keep the first 8192 / 16384 in LDS and the rest in HBM, so a cell in
    which (almost) every pixel is a candidate - a periodic texture, dense noise - gives the oracle's lists instead of an error, and
    the calls that follow are unaffected."""
    w, h = 300, 280
    tile = np.array([[10, 120, 60], [100, 5, 90], [40, 110, 20]], np.uint8)   # period-3 texture: tens of thousands of equal responses
    periodic = np.tile(tile, (h // 3 + 1, w // 3 + 1))[:h, :w].copy()
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    plateau = np.kron(rng.integers(0, 2, (h // 2, w // 2)).astype(np.uint8) * 200 + 20, np.ones((2, 2), np.uint8))   # 2x2 blocks: plateaus of equal maxima
    scene = _frames(pmv, KITTI07, 1, seed=1001)[0][:h, :w].copy()
    ctx = gpu_ctx_factory(w, h, n_slots=1)
    cells = pmv.grid_cells(w, h)
    seen_big = 0
    for img in (periodic, noise, plateau, scene):
        ctx.frame_upload(0, img)
        for c, (gxy, gsc) in zip(cells, ctx.detect_shitomasi(0, cells, 40)):
            rxy, rsc, R = orc.shitomasi_cell(img, c, 40, want_resp=True)
            R = np.nan_to_num(R)
            seen_big += int((R > 0.4 * R.max()).sum() > 8192)
            assert np.array_equal(gxy, rxy) and np.array_equal(gsc, rsc)
        for mx, md in ((40, 5.0), (300, 1.0), (60, 0.5)):
            for c, g in zip(cells, ctx.detect_gftt(0, cells, mx, min_dist=md)):
                assert np.array_equal(g, orc.gftt_cell(img, c, mx, min_dist=md))
    assert seen_big >= 2, "the inputs were meant to exceed the LDS candidate capacity"


def test_shitomasi_matches_oracle(pmv, orc, gpu_ctx_factory):
    cfg = KITTI07
    fr = _frames(pmv, cfg, 1, seed=1001)[0]
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=1)
    ctx.frame_upload(0, fr)
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    got = ctx.detect_shitomasi(0, cells, 40)
    for c, (gxy, gsc) in zip(cells, got):
        rxy, rsc, R = orc.shitomasi_cell(fr, c, 40, want_resp=True)
        Rg = ctx.shitomasi_response(0, c)
        assert np.array_equal(np.isnan(Rg), np.isnan(R))
        assert np.array_equal(np.nan_to_num(Rg), np.nan_to_num(R)), "response differs"
        assert np.array_equal(gxy, rxy) and np.array_equal(gsc, rsc)


def _check_lk(ctx, orc, a, b, pts):
    xy, st, err = ctx.lk_track(0, 1, pts)
    rxy, rst, rerr, _ = orc.lk_track(a, b, pts)
    assert np.array_equal(st, rst)
    ok = st > 0
    assert np.array_equal(xy[ok], rxy[ok])       # float32 positions, bit-exact
    assert np.array_equal(err[ok], rerr[ok])
    # what the reference keeps: Feature(int(x), int(y)) for status==1 (OpenCVLucasKanadeFM.cpp:25)
    assert np.array_equal(xy[ok].astype(np.int32), rxy[ok].astype(np.int32))
    return ok


@pytest.mark.parametrize("cfg,per_cell", [(KITTI07, 40), (KITTI00, 80)])
def test_lk_matches_oracle_on_sequence(pmv, orc, gpu_ctx_factory, cfg, per_cell):
    fr = _frames(pmv, cfg, 2, seed=1007, first=10)
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=2)
    ctx.frame_upload(0, fr[0])
    ctx.frame_upload(1, fr[1])
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    pts = np.concatenate([d + c[:2] for c, d in zip(cells, ctx.detect_gftt(0, cells, per_cell))]).astype(np.float32)
    ok = _check_lk(ctx, orc, fr[0], fr[1], pts)
    assert ok.sum() > 0.6 * len(pts)


def test_lk_edge_cases(pmv, orc, gpu_ctx_factory):
    cfg = KITTI07
    fr = _frames(pmv, cfg, 2, seed=1002, first=3)
    w, h = cfg["w"], cfg["h"]
    ctx = gpu_ctx_factory(w, h, n_slots=2)
    ctx.frame_upload(0, fr[0])
    ctx.frame_upload(1, fr[1])
    rng = np.random.default_rng(11)
    border = np.array([[0, 0], [w - 1, h - 1], [-20.0, 5], [w + 10.0, h + 3.0], [-40, -40], [3, h - 2], [w - 2, 1],
                       [w + 200.0, 10], [15.5, 15.5], [w - 16.5, h - 16.5]], np.float32)
    rnd = np.stack([rng.uniform(-30, w + 30, 400), rng.uniform(-30, h + 30, 400)], 1).astype(np.float32)
    _check_lk(ctx, orc, fr[0], fr[1], np.concatenate([border, rnd]))
    # empty input
    xy, st, err = ctx.lk_track(0, 1, np.zeros((0, 2), np.float32))
    assert len(xy) == 0
    # textureless image: every track is rejected by the min-eigenvalue test at level 0
    flat = np.full((h, w), 100, np.uint8)
    ctx.frame_upload(0, flat)
    ctx.frame_upload(1, flat)
    ok = _check_lk(ctx, orc, flat, flat, rnd[:50])
    assert ok.sum() == 0


def test_lk_recovers_known_translation(pmv, gpu_ctx_factory):
    """KA7: pure integer translation of a textured image is recovered to sub-pixel accuracy (size-independent property)."""
    cfg = KITTI00
    base = _frames(pmv, cfg, 1, seed=1009, first=5)[0]
    dx, dy = 7, -3
    shifted = np.roll(np.roll(base, dy, 0), dx, 1)
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=2)
    ctx.frame_upload(0, base)
    ctx.frame_upload(1, shifted)
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    pts = np.concatenate([d + c[:2] for c, d in zip(cells, ctx.detect_gftt(0, cells, 40))]).astype(np.float32)
    inner = (pts[:, 0] > 60) & (pts[:, 0] < cfg["w"] - 60) & (pts[:, 1] > 60) & (pts[:, 1] < cfg["h"] - 60)
    xy, st, _ = ctx.lk_track(0, 1, pts[inner])
    d = xy[st > 0] - pts[inner][st > 0]
    assert (st > 0).mean() > 0.95
    assert np.abs(np.median(d, 0) - [dx, dy]).max() < 0.05


def test_error_paths(pmv, gpu_ctx_factory):
    ctx = gpu_ctx_factory(200, 100, n_slots=1, max_tracks=8)
    with pytest.raises(pmv.PmvError):
        ctx.frame_upload(1, np.zeros((100, 200), np.uint8))          # slot out of range
    with pytest.raises(pmv.PmvError):
        ctx.frame_upload(0, np.zeros((101, 200), np.uint8))          # larger than capacity
    with pytest.raises(pmv.PmvError):
        ctx.lk_track(0, 0, np.zeros((4, 2), np.float32))             # no pyramid yet
    ctx.frame_upload(0, np.zeros((100, 200), np.uint8))
    with pytest.raises(pmv.PmvError):
        ctx.lk_track(0, 0, np.zeros((9, 2), np.float32))             # exceeds max_tracks
    with pytest.raises(pmv.PmvError):
        ctx.detect_gftt(0, np.asarray([[0, 0, 256, 50]], np.int32), 10)   # cell wider than 255


def test_committed_golden_vectors(pmv, gpu_ctx_factory):
    """tests/golden/frontend_640x200.npz (made by tests/golden/make_fixtures.py with the oracle): corner lists, LK positions,
    status and error must be reproduced bit for bit from the regenerated seeded frames."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_640x200.npz"))
    w, h, f = int(g["w"]), int(g["h"]), float(g["f"])
    frames, _ = pmv.synth_sequence(int(g["seed"]), int(g["first"]), 2, w, h, f, f, 320.0, 100.0, nthreads=4)
    ctx = gpu_ctx_factory(w, h, n_slots=2)
    ctx.frame_upload(0, frames[0])
    ctx.frame_upload(1, frames[1])
    cells = pmv.grid_cells(w, h)
    det = ctx.detect_gftt(0, cells, 20)
    assert [len(d) for d in det] == list(g["corner_counts"])
    assert np.array_equal(np.concatenate(det), g["corners"].astype(np.int32))
    pts = np.concatenate([d + c[:2] for c, d in zip(cells, det)]).astype(np.float32)
    xy, st, err = ctx.lk_track(0, 1, pts)
    assert ctx.num_levels(0) == int(g["levels"])
    assert np.array_equal(st, g["lk_status"])
    ok = st > 0
    assert np.array_equal(xy[ok], g["lk_xy"][ok]) and np.array_equal(err[ok], g["lk_err"][ok])


def test_knn_matcher_matches_oracle(pmv, orc, gpu_ctx_factory):
    """kNNFeatureMatcher's arithmetic (kNNFeatureMatcher.cpp:13-31, 63-122) on the GPU: best-fit indices and window errors
    bit-exact against the oracle (which is pinned by twins written from the reference source), incl. the quirks: fewer candidates
    than neighbours, no candidate at all, candidates at the source's own coordinates, windows cut by the border, identical images."""
    cfg = KITTI07
    fr = _frames(pmv, cfg, 2, seed=1003)
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=2, max_tracks=2048)
    ctx.frame_upload(0, fr[0]); ctx.frame_upload(1, fr[1])
    rng = np.random.default_rng(3)
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    src = np.concatenate([orc.gftt_cell(fr[0], c, 40) + c[:2] for c in cells])
    for m in (1000, 300, 5, 0):
        cmp_xy = np.unique(np.stack([rng.integers(0, cfg["w"], m), rng.integers(0, cfg["h"], m)], 1), axis=0) if m else np.zeros((0, 2), np.int32)
        rng.shuffle(cmp_xy)
        if m >= 300:
            cmp_xy[:50] = src[:50] + rng.integers(-3, 4, (50, 2))        # plausible matches close by
            cmp_xy[50] = src[60]                                          # same coordinates as a source feature
        s2 = np.concatenate([src, [[0, 0], [cfg["w"] - 1, cfg["h"] - 1], [3, cfg["h"] - 2]]])
        for a, b, sa, sb in ((fr[0], fr[1], 0, 1), (fr[0], fr[0], 0, 0)):
            gb, ge = ctx.knn_match(sa, sb, s2, cmp_xy)
            ob_, oe = orc.knn_match(a, b, s2, cmp_xy)
            assert np.array_equal(gb, ob_) and np.array_equal(ge, oe)


def test_fast_extractor_matches_oracle(pmv, orc, gpu_ctx_factory):
    """cv::FAST 9_16 restatement: keypoint lists (raster order), responses and the first-`max` cut bit-exact per grid cell, on a whole
    frame (the kNN matcher's call), with and without non-maximum suppression, at several thresholds, on noise and on tiny views"""
    cfg = KITTI00
    fr = _frames(pmv, cfg, 1, seed=1001)[0]
    rng = np.random.default_rng(8)
    noise = rng.integers(0, 256, fr.shape, dtype=np.uint8)
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=1)
    cells = pmv.grid_cells(cfg["w"], cfg["h"])
    whole = np.asarray([[0, 0, cfg["w"], cfg["h"]]], np.int32)
    tiny = np.asarray([[5, 7, 6, 9], [10, 10, 7, 7], [0, 0, 12, 8]], np.int32)
    for img in (fr, noise):
        ctx.frame_upload(0, img)
        for t, nonmax, mx in ((10, True, 40), (10, False, 500), (30, True, 1000), (0, True, 64)):
            for views in (cells, whole, tiny):
                got = ctx.detect_fast(0, views, mx, threshold=t, nonmax=nonmax)
                for c, (gxy, grs) in zip(views, got):
                    rxy, rrs = orc.fast9_cell(img, c, mx, threshold=t, nonmax=nonmax)
                    assert np.array_equal(gxy, rxy) and np.array_equal(grs, rrs), (t, nonmax, mx, c)
        assert sum(len(xy) for xy, _ in ctx.detect_fast(0, whole, 1000)) > 100
    assert all(len(xy) == 0 for xy, _ in ctx.detect_fast(0, cells, 0))


def test_lk_work_counters_are_summed_from_the_per_track_words(pmv, gpu_ctx_factory):
    """pmv_lk_counters: iterations / level passes / tracks of the launches since the last reset (bench.py's OPS_lk). The kernels write one 16-bit
    word per track next to the results and the host sums them (three atomics per track on shared counters cost the batched leg 14 %)."""
    cfg = dict(w=320, h=240, fx=300.0, fy=300.0, cx=160.0, cy=120.0)
    frames, _ = pmv.synth_sequence(5, 0, 3, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=4)
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=3)
    ctx.frames_stage(0, frames); ctx.frames_build(0, 3)
    pts = np.concatenate([c for c in ctx.detect_gftt(0, pmv.grid_cells(cfg["w"], cfg["h"]), 20) if len(c)]).astype(np.float32)
    assert len(pts) > 20
    ctx.lk_counters(reset=True)
    ctx.lk_track(0, 1, pts)
    it1, lv1, n1 = ctx.lk_counters()
    assert n1 == len(pts) and 0 < lv1 <= 5 * len(pts) and lv1 <= it1 <= 30 * lv1
    ctx.lk_track(1, 2, pts[:10])
    it2, lv2, n2 = ctx.lk_counters(reset=True)
    assert n2 == len(pts) + 10 and it2 > it1 and lv2 > lv1
    assert tuple(ctx.lk_counters()) == (0, 0, 0)
