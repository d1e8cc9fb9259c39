"""The host orchestration (addFrame, initialise, estimatePose and the gather/scatter halves of the five adapters) is ONE piece of
code, practical-multi-view_amd/host/vo_*.cpp, compiled into both the product and the oracle - so "GPU pipeline == oracle pipeline"
says nothing about it. Here it is checked against tests/twin/ref_twin.cpp: a second restatement written separately from the
reference's sources (own Feature / Frame / Feature3D with the reference's containers, Frame copies + write-back as
OdometryPipeline.cpp:237-243,400-401 do, std::find erases, std::map tr_opt ...), which includes nothing from host/ and calls only the
oracle's LEAF functions (orc_gftt_cell, orc_lk_track, orc_pnp_ransac, orc_ba_solve, the two-view leaves).

Bar: every frame's features - (column, row, landmark id) in container iteration order - and the feat_corr sizes (quirk Q10's
empty entries) identical, poses to 1e-12. Quirk coverage: each `variant` of the twin switches ONE quirk off; on the fixtures it
must then differ from the shared orchestration, while the faithful twin does not - i.e. the fixture exercises the quirk and both
implementations have it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TW = os.path.join(ROOT, "tests", "twin")
_u8p, _i32p, _f64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def twin():
    ob.load()                                                                  # builds oracle/liborc.so if needed
    C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"), mode=C.RTLD_GLOBAL)      # the leaves the twin resolves at load time
    so, src = os.path.join(TW, "libref_twin.so"), os.path.join(TW, "ref_twin.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        # no -I: the twin must compile without a single header of this repo; same FP flags as the oracle
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", src, "-o", so])
    lib = C.CDLL(so)
    lib.twin_run.restype = C.c_void_p
    lib.twin_run.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _f64p, _f64p] + [C.c_int] * 7
    for f in ("twin_free", "twin_num_poses", "twin_num_frames"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.twin_get_poses.argtypes = [C.c_void_p, _f64p]
    lib.twin_frame_feature_count.argtypes = [C.c_void_p, C.c_int]
    lib.twin_frame_corr_count.argtypes = [C.c_void_p, C.c_int]
    lib.twin_get_frame_features.argtypes = [C.c_void_p, C.c_int, _i32p]
    lib.twin_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    return lib


class TwinResult:
    def __init__(self, lib, h):
        n = lib.twin_num_poses(h)
        self.poses = np.zeros((n, 12))
        lib.twin_get_poses(h, self.poses.ctypes.data_as(_f64p))
        self.features, self.corr_counts = [], []
        for k in range(lib.twin_num_frames(h)):
            c = lib.twin_frame_feature_count(h, k)
            a = np.zeros((c, 3), np.int32)
            if c:
                lib.twin_get_frame_features(h, k, a.ctypes.data_as(_i32p))
            self.features.append(a)
            self.corr_counts.append(lib.twin_frame_corr_count(h, k))
        cnt = (C.c_longlong * 10)()
        lib.twin_get_counters(h, cnt)
        keys = ("init_offset", "n_landmarks", "scale_e6", "q4_effects", "q10_inserts", "erased", "heuristic", "pnp_calls", "tri_calls", "ba_calls")
        self.counters = dict(zip(keys, [int(v) for v in cnt]))
        lib.twin_free(h)


def run_twin(lib, frames, K, gt, min_tracked=400, tol=150, init_frames=5, bundle_size=5, ba_iterations=5, extractor=0, variant=0):
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w = frames.shape
    Kd = np.ascontiguousarray(K, np.float64).reshape(9)
    g = np.ascontiguousarray(gt, np.float64).reshape(n, 12)
    hnd = lib.twin_run(frames.ctypes.data_as(_u8p), n, w, h, Kd.ctypes.data_as(_f64p), g.ctypes.data_as(_f64p), min_tracked, tol, init_frames,
                       bundle_size, ba_iterations, extractor, variant)
    return TwinResult(lib, hnd)


def same(t, o):
    """features (column, row, landmark id) in iteration order, feat_corr sizes and poses all equal"""
    if len(t.features) != len(o.features) or t.poses.shape != o.poses.shape:
        return False
    return all(np.array_equal(a, b) for a, b in zip(t.features, o.features)) and list(t.corr_counts) == list(o.corr_counts) and \
        float(np.abs(t.poses - o.poses).max()) <= 1e-12


def assert_same(t, o):
    assert len(t.features) == len(o.features)
    for k, (a, b) in enumerate(zip(t.features, o.features)):
        assert np.array_equal(a, b), f"frame {k}: features / landmark ids / container order differ"
    assert list(t.corr_counts) == list(o.corr_counts), "feat_corr sizes differ (quirk Q10's insertions)"
    assert t.poses.shape == o.poses.shape and np.abs(t.poses - o.poses).max() <= 1e-12
    assert t.counters["init_offset"] == int(o.stats["init_offset"]) and t.counters["n_landmarks"] == int(o.stats["n_landmarks"])
    assert abs(t.counters["scale_e6"] - o.stats["scale"] * 1e6) <= 1
    for a, b in (("pnp_calls", "pnp_calls"), ("tri_calls", "tri_calls"), ("ba_calls", "ba_calls"), ("heuristic", "heuristic_motion")):
        assert t.counters[a] == int(o.stats[b]), a


def small_case(pmv):
    g = np.load(os.path.join(ROOT, "tests", "golden", "pipeline_small.npz"))
    w, h, f, n, seed = int(g["w"]), int(g["h"]), float(g["f"]), int(g["n"]), int(g["seed"])
    frames, gt = pmv.synth_sequence(seed, 0, n, w, h, f, f, w / 2.0, h / 2.0, nthreads=8)
    K = np.array([f, 0, w / 2.0, 0, f, h / 2.0, 0, 0, 1.0])
    return frames, K, gt, dict(min_tracked=200, tol=75, bundle_size=3), g


def test_twin_reproduces_the_golden_small_pipeline(pmv, twin):
    frames, K, gt, kw, g = small_case(pmv)
    for threaded in (0, 1):   # the shared code's one-thread schedule and its front-end / back-end threads
        o = ob.run_pipeline(frames, K, gt, threaded=threaded, n_threads=4, **kw)
        t = run_twin(twin, frames, K, gt, **kw)
        assert_same(t, o)
    # and the committed fixture itself (generated in an earlier round by the shared code): the twin meets it on its own
    assert np.array_equal(np.array([len(a) for a in t.features]), g["counts"])
    assert np.array_equal(np.concatenate([a[:, :2] for a in t.features]).astype(np.int16), g["features"])
    assert np.abs(t.poses - g["poses"]).max() <= 1e-12
    print("twin counters (small fixture):", t.counters)


K00 = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)


@pytest.fixture(scope="module")
def metric_prefix(pmv):
    c = K00
    frames, gt = pmv.synth_sequence(1007, 0, 60, c["w"], c["h"], c["fx"], c["fy"], c["cx"], c["cy"], nthreads=8)
    return frames, np.array([c["fx"], 0, c["cx"], 0, c["fy"], c["cy"], 0, 0, 1.0]), gt


def test_twin_reproduces_the_metric_config_prefix(pmv, twin, metric_prefix):
    """the first 60 frames of BASELINE configs[1] (1241x376, 400 tracks, tol 150, bundle 5): PnP with outlier erases, BA every second
    frame, re-detections, triangulation fallbacks"""
    frames, K, gt = metric_prefix
    o = ob.run_pipeline(frames, K, gt, threaded=1, n_threads=8)
    t = run_twin(twin, frames, K, gt)
    assert_same(t, o)
    c = t.counters
    print("twin counters (metric prefix):", c)
    assert c["pnp_calls"] > 20 and c["ba_calls"] > 20 and c["erased"] > 0 and c["q10_inserts"] > 0


# variant bit -> the quirk it switches off
QUIRKS = {1: "Q3 re-detection on the previous frame's image", 2: "Q4 hasNeighbor on cell-local coordinates", 8: "RANSAC outliers erased from feats3d",
          16: "Q10 operator[] inserts empty feat_corr entries", 32: "Q7 float32 round trip of the landmarks in place"}


@pytest.mark.parametrize("bit", sorted(QUIRKS))
def test_quirks_are_exercised_and_shared(pmv, twin, metric_prefix, bit):
    frames, K, gt = metric_prefix
    n = 40
    o = ob.run_pipeline(frames[:n], K, gt[:n], threaded=0, n_threads=4)
    assert same(run_twin(twin, frames[:n], K, gt[:n]), o)
    v = run_twin(twin, frames[:n], K, gt[:n], variant=bit)
    assert not same(v, o), f"the fixture does not exercise: {QUIRKS[bit]}"
    if bit == 16:   # Q10 changes nothing but the size of the written-back feat_corr copies (features and poses stay)
        assert all(np.array_equal(a, b) for a, b in zip(v.features, o.features)) and np.abs(v.poses - o.poses).max() <= 1e-12
        assert sum(o.corr_counts) - sum(v.corr_counts) == run_twin(twin, frames[:n], K, gt[:n]).counters["q10_inserts"] > 0
    if bit == 2:
        assert run_twin(twin, frames[:n], K, gt[:n]).counters["q4_effects"] > 0


def test_initialise_quirks_q5_q6(pmv, twin):
    """Q6: initialise asks for min_tracked / cells features per cell in INTEGER arithmetic (405 / 10 = 40) where addFrame uses
    ceil (41). Q5: the cost of an init frame is std(count per cell) + std(score); OpenCVGoodFeatureExtractor leaves score = 0, so
    only an extractor that sets it (ShiTomasi) makes the second term matter - and may pick another start frame."""
    c = K00
    frames, gt = pmv.synth_sequence(1003, 0, 16, c["w"], c["h"], c["fx"], c["fy"], c["cx"], c["cy"], nthreads=8)
    K = np.array([c["fx"], 0, c["cx"], 0, c["fy"], c["cy"], 0, 0, 1.0])
    kw = dict(min_tracked=405, tol=150, bundle_size=5)
    o = ob.run_pipeline(frames, K, gt, threaded=0, **kw)
    assert same(run_twin(twin, frames, K, gt, **kw), o)
    assert len(o.features[0]) <= 400                                   # 10 cells x 40, not x 41
    assert not same(run_twin(twin, frames, K, gt, variant=4, **kw), o)
    # Q5 with ShiTomasi scores: faithful twin equal; without the score term the chosen start frame (or nothing) may change - the
    # test only demands that the faithful form is the shared code's
    for seed in (1003, 1004, 1005):
        fr, g = pmv.synth_sequence(seed, 0, 12, 620, 188, 355.0, 355.0, 310.0, 94.0, nthreads=8)
        K2 = np.array([355.0, 0, 310.0, 0, 355.0, 94.0, 0, 0, 1.0])
        kw2 = dict(min_tracked=200, tol=75, bundle_size=3, extractor=1)
        o2 = ob.run_pipeline(fr, K2, g, threaded=0, **kw2)
        t2 = run_twin(twin, fr, K2, g, **kw2)
        assert_same(t2, o2)
        v2 = run_twin(twin, fr, K2, g, variant=64, **kw2)
        print(f"seed {seed}: init_offset with score term {t2.counters['init_offset']}, without {v2.counters['init_offset']}")


def test_retired_frames_export_what_live_frames_would(pmv, twin):
    """The shared orchestration retires frames that left the bundle window (their tables are recycled, exports come from a compact copy).
    Without bundle adjustment (bundle_size 0) the window is the shortest there is - every frame but the last few is retired during the run -
    and the exports still equal the twin's, which keeps every frame in the reference's own containers; both schedules."""
    frames, K, gt, kw, _ = small_case(pmv)
    kw = dict(kw, bundle_size=0)
    t = run_twin(twin, frames, K, gt, **kw)
    for threaded in (0, 1):
        o = ob.run_pipeline(frames, K, gt, threaded=threaded, n_threads=2, **kw)
        assert_same(t, o)
    assert len(t.features) == len(frames) - t.counters["init_offset"] if "init_offset" in t.counters else True
