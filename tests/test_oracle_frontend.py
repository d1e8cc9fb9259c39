"""Oracle (CPU restatement) front-end against independent numpy restatements and hand-computable known answers.
The reference ships no fixtures, so what can be pinned is: in-repo arithmetic (ShiTomasi, quirk Q1) exactly,
and the published OpenCV algorithms against a second, independently written implementation (parity with OpenCV's bits stays
UNPINNED — oracle/orc_common.h)."""
import numpy as np
import pytest


def _reflect101(p, n):
    p = np.asarray(p)
    p = np.where(p < 0, -p, p)
    return np.where(p >= n, 2 * n - 2 - p, p)


def np_pyr_down(img):
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)
    ys = _reflect101(2 * np.arange(dh)[:, None] + np.arange(-2, 3)[None, :], h)
    xs = _reflect101(2 * np.arange(dw)[:, None] + np.arange(-2, 3)[None, :], w)
    a = img.astype(np.int64)[ys]                 # dh,5,w
    rows = np.tensordot(k, a, axes=([0], [1]))   # dh,w
    cols = rows[:, xs]                           # dh,dw,5
    return ((cols @ k + 128) >> 8).astype(np.uint8)


def np_scharr(img):
    a = np.pad(img.astype(np.int64), 1, mode="reflect")
    t0 = (a[:-2] + a[2:]) * 3 + a[1:-1] * 10          # vertical smoothing, h x (w+2)
    t1 = a[2:] - a[:-2]
    dx = t0[:, 2:] - t0[:, :-2]
    dy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return np.stack([dx, dy], -1).astype(np.int16)


@pytest.mark.parametrize("shape", [(37, 53), (94, 311), (188, 621), (33, 34)])
def test_pyr_down_matches_numpy(orc, shape):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(orc.pyr_down(img), np_pyr_down(img))


def test_pyr_down_constant_and_size_rule(orc):
    img = np.full((47, 156), 201, np.uint8)
    out = orc.pyr_down(img)
    assert out.shape == (24, 78) and (out == 201).all()


@pytest.mark.parametrize("shape", [(40, 60), (47, 156), (376, 1241)])
def test_scharr_matches_numpy(orc, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(orc.scharr(img), np_scharr(img))


def test_lk_level_count_rule(orc, pmv):
    """buildOpticalFlowPyramid stops when the NEXT level would be <= winSize (SURVEY.md §8 header): KITTI -> levels 0..3."""
    for (w, h), want in (((1241, 376), 3), ((1226, 370), 3), ((1920, 1080), 4), ((200, 100), 1), ((70, 70), 1), ((64, 64), 0)):
        img = np.zeros((h, w), np.uint8)
        _, _, _, lv = orc.lk_track(img, img, np.zeros((1, 2), np.float32))
        assert lv == want, (w, h, lv)


def _texture(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2)).astype(np.float64)
    img = np.kron(base, np.ones((4, 4)))[:h, :w]
    # smooth a little so that LK's linearisation holds
    k = np.array([1, 4, 6, 4, 1], float) / 16
    img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, img)
    img = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, img)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_lk_recovers_integer_translation(orc):
    """KA7: a pure (dx,dy) shift of a textured image is recovered to a few hundredths of a pixel."""
    a = _texture(240, 400, 3)
    dx, dy = 5, -4
    b = np.roll(np.roll(a, dy, 0), dx, 1)
    ys, xs = np.mgrid[60:180:20, 60:340:20]
    pts = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.float32)
    out, st, err, lv = orc.lk_track(a, b, pts)
    assert st.all()
    d = out - pts
    assert np.abs(d - [dx, dy]).max() < 0.1
    # identical images: zero flow, zero error
    out, st, err, _ = orc.lk_track(a, a, pts)
    assert st.all() and np.abs(out - pts).max() < 1e-3 and err.max() == 0


def test_lk_status_rules(orc):
    a = _texture(120, 200, 5)
    flat = np.full_like(a, 90)
    pts = np.array([[100, 60], [-50, 10], [260, 60], [100, -40]], np.float32)
    _, st, _, _ = orc.lk_track(flat, flat, pts[:1])
    assert st[0] == 0                                  # min-eigenvalue test at level 0
    _, st, _, _ = orc.lk_track(a, a, pts)
    assert st[0] == 1 and not st[1:].any()             # windows starting outside [-win, size) fail
    out, st, err, _ = orc.lk_track(a, a, np.zeros((0, 2), np.float32))
    assert len(st) == 0


@pytest.mark.parametrize("shape", [(370, 1226), (47, 156), (33, 40), (101, 77), (64, 35)])
def test_fast_pyr_down_twin_is_bit_identical(orc, shape):
    img = np.random.default_rng(shape[0]).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(orc.pyr_down_fast(img), orc.pyr_down(img))


def test_fast_lk_twin_is_bit_identical(orc):
    """orc_fast.cpp (padded buffers, vectorisable row loops, thread pool) is the cpu_baseline's LK: same positions, status
    and err as the plain restatement, for interior, border, outside, flat and non-converging tracks, any thread count."""
    rng = np.random.default_rng(11)
    for (h, w), seed in (((240, 400), 3), ((120, 200), 5), ((70, 66), 9)):
        a = _texture(h, w, seed)
        b = np.roll(np.roll(a, -3, 0), 4, 1)
        b[: h // 3] = rng.integers(0, 256, (h // 3, w), dtype=np.uint8)           # a third of the image unrelated: non-convergence
        pts = np.concatenate([rng.uniform(-45, max(w, h) + 45, (300, 2)), [[0, 0], [w - 1, h - 1], [-32.4, 5], [w - 0.6, h - 0.6], [3.5, h + 31.4]]]).astype(np.float32)
        ref = orc.lk_track(a, b, pts)
        for nt in (1, 3, 8):
            got = orc.lk_track_fast(a, b, pts, nt)
            for x, y in zip(ref[:3], got):
                assert np.array_equal(x, y)
        flat = np.full_like(a, 90)
        ref = orc.lk_track(flat, b, pts)
        got = orc.lk_track_fast(flat, b, pts, 2)
        for x, y in zip(ref[:3], got):
            assert np.array_equal(x, y)
    got = orc.lk_track_fast(a, b, np.zeros((0, 2), np.float32), 2)
    assert len(got[1]) == 0


def _corner_image(h, w, step=24):
    img = np.full((h, w), 40, np.uint8)
    for y in range(10, h - 10, step):
        for x in range(10, w - 10, step):
            img[y:y + step // 2, x:x + step // 2] = 200
    return img


def test_gftt_finds_checker_corners_in_order_with_min_distance(orc):
    """KA7: responses peak at the block corners; output respects min distance and descending response."""
    img = _corner_image(120, 160)
    cell = (0, 0, 160, 120)
    xy, eig = orc.gftt_cell(img, cell, 500, want_eig=True)
    assert len(xy) > 20
    vals = eig[xy[:, 1], xy[:, 0]]
    assert (np.diff(vals) <= 0).all(), "corners are returned in descending response order"
    d = np.linalg.norm(xy[:, None, :].astype(float) - xy[None, :, :], axis=2)
    d[np.arange(len(xy)), np.arange(len(xy))] = 99
    assert d.min() >= 5.0
    assert ((xy[:, 0] >= 1) & (xy[:, 0] < 159) & (xy[:, 1] >= 1) & (xy[:, 1] < 119)).all()
    # every returned corner is within 2 px of a true block corner
    cy = np.array(sorted(set(list(range(10, 110, 24)) + [v + 12 for v in range(10, 110, 24)])))
    cx = np.array(sorted(set(list(range(10, 150, 24)) + [v + 12 for v in range(10, 150, 24)])))
    assert (np.abs(xy[:, 0][:, None] - cx[None]).min(1) <= 2).all() and (np.abs(xy[:, 1][:, None] - cy[None]).min(1) <= 2).all()
    # max_corners truncates the same list
    assert np.array_equal(orc.gftt_cell(img, cell, 7), xy[:7])


def test_gftt_flat_image_and_float32_response(orc):
    flat = np.full((100, 100), 9, np.uint8)
    xy, eig = orc.gftt_cell(flat, (0, 0, 100, 100), 50, want_eig=True)
    assert len(xy) == 0 and (eig == 0).all()
    # vertical step edge: min eigenvalue ~ 0 everywhere (one-dimensional structure)
    edge = np.zeros((60, 60), np.uint8)
    edge[:, 30:] = 255
    _, eig = orc.gftt_cell(edge, (0, 0, 60, 60), 5, want_eig=True)
    assert np.abs(eig).max() < 1e-6
    assert eig.dtype == np.float32


def test_gftt_sobel_reads_across_cell_borders_but_box_reflects(orc):
    """cell = non-isolated ROI: pixels just outside the cell change Sobel at the cell edge."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (90, 90), dtype=np.uint8)
    img2 = img.copy()
    img2[:, 45] = 255 - img2[:, 45]          # column just right of the cell (0,0,45,90)
    _, e1 = orc.gftt_cell(img, (0, 0, 45, 90), 5, want_eig=True)
    _, e2 = orc.gftt_cell(img2, (0, 0, 45, 90), 5, want_eig=True)
    assert not np.array_equal(e1[:, -2:], e2[:, -2:])      # last columns see the neighbour through Sobel (and the box sum)
    assert np.array_equal(e1[:, :-3], e2[:, :-3])


def np_shitomasi(cell):
    """ShiTomasiFeatureExtractor.cpp:49-75 on Frame.cpp:58-86 / :119-138, written independently with numpy."""
    s = cell.view(np.int8).astype(np.float64)          # quirk Q1: gray bytes read as SIGNED char
    h, w = s.shape
    gx = np.zeros((h, w))
    gy = np.zeros((h, w))
    gx[1:-1, 1:-1] = 0.5 * s[1:-1, 2:] - 0.5 * s[1:-1, :-2]
    gy[1:-1, 1:-1] = 0.5 * s[2:, 1:-1] - 0.5 * s[:-2, 1:-1]
    H = np.stack([gx * gx, gy * gy, gx * gy], -1)
    P = np.pad(H, ((1, 1), (1, 1), (0, 0)), mode="reflect")
    B = np.zeros_like(H)
    for j in range(3):
        for i in range(3):
            B += P[j:j + h, i:i + w]
    B *= 1.0 / 9.0
    Ixx, Iyy, Ixy = B[..., 0], B[..., 1], B[..., 2]
    Bq = -Ixx - Iyy
    Cq = Ixx * Iyy - Ixy * Ixy
    with np.errstate(invalid="ignore"):
        disc = np.sqrt(Bq * Bq - 4 * Cq)
    l1, l2 = (-Bq + disc) / 2, (-Bq - disc) / 2
    R = np.where(l2 < l1, l2, l1)
    R[:, -1] = 0.0                                       # last column is skipped (:58)
    return R


@pytest.mark.parametrize("seed", [1, 2])
def test_shitomasi_matches_in_repo_arithmetic(orc, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (80, 110), dtype=np.uint8)
    cell = (7, 5, 90, 60)
    xy, sc, R = orc.shitomasi_cell(img, cell, 25, want_resp=True)
    ref = np_shitomasi(img[5:65, 7:97])
    assert np.array_equal(np.isnan(R), np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(R), np.nan_to_num(ref), rtol=1e-13, atol=1e-12)
    # threshold 0.4*max, sorted by score, ties in raster order, at most `max`
    thr = np.nanmax(ref) * 0.4
    cand = [(-(ref[y, x]), y * 90 + x) for y in range(60) for x in range(90) if ref[y, x] > thr]
    cand.sort()
    want = np.array([[i % 90, i // 90] for _, i in cand[:25]], np.int32)
    assert np.array_equal(xy, want)
    assert (np.diff(sc) <= 0).all()


def test_shitomasi_signed_char_quirk_q1(orc):
    """KA4: the same edge gives a different response once pixel values cross 127 (they wrap negative)."""
    lo = np.zeros((40, 40), np.uint8); lo[20:, 20:] = 100     # 0 -> 100: gradient +50
    hi = np.zeros((40, 40), np.uint8); hi[20:, 20:] = 200     # 0 -> 200 reads as 0 -> -56: gradient -28
    _, _, Rl = orc.shitomasi_cell(lo, (0, 0, 40, 40), 5, want_resp=True)
    _, _, Rh = orc.shitomasi_cell(hi, (0, 0, 40, 40), 5, want_resp=True)
    assert np.nanmax(Rh) < np.nanmax(Rl)
    np.testing.assert_allclose(np.nanmax(Rh) / np.nanmax(Rl), (56.0 / 100.0) ** 2, rtol=1e-9)


def test_bgr2gray_matches_numpy_and_is_identity_on_gray(orc):
    """Frame::init's cvtColor(BGR2GRAY) (Frame.cpp:40-41): the oracle against a numpy twin of the 14-bit fixed-point formula; a gray
    image stored as BGR (B = G = R: what imread(IMREAD_COLOR) makes of KITTI's PNGs) comes back unchanged (quirk Q2)."""
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, (57, 83, 3), dtype=np.uint8)
    twin = ((bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 + bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(orc.bgr2gray(bgr), twin)
    g = rng.integers(0, 256, (40, 61), dtype=np.uint8)
    assert np.array_equal(orc.bgr2gray(np.repeat(g[..., None], 3, axis=2)), g)
    ramp = np.arange(256, dtype=np.uint8)
    assert np.array_equal(orc.bgr2gray(np.stack([ramp, ramp, ramp], -1)[None]), ramp[None])
