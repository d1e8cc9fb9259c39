"""KITTI file formats and the reference's error report — host-side, CPU only."""
import importlib
import os

import numpy as np

pmv = importlib.import_module("practical-multi-view_amd")
kitti = importlib.import_module("practical-multi-view_amd.kitti")


def test_calibration_line_parser_follows_the_reference(tmp_path):
    p = tmp_path / "calib.txt"
    p.write_text("P0: 7.188560000000e+02 0.000000000000e+00 6.071928000000e+02 0.000000000000e+00 0.000000000000e+00 "
                 "7.188560000000e+02 1.852157000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 "
                 "1.000000000000e+00 0.000000000000e+00\nP1: 1 2 3 4 5 6 7 8 9 10 11 12\n")
    K = kitti.parse_calibration(str(p), 0)
    np.testing.assert_allclose(K, [[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]])
    K1 = kitti.parse_calibration(str(p), 1)
    np.testing.assert_allclose(K1, [[1, 2, 3], [5, 6, 7], [9, 10, 11]])   # 4th column skipped; the last token is never read


def test_pose_round_trip_and_stop(tmp_path):
    rng = np.random.default_rng(3)
    P = rng.normal(size=(7, 12))
    f = tmp_path / "07.txt"
    kitti.write_poses_kitti(str(f), np.hstack([P[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]], P[:, [3, 7, 11]]]))
    Q = kitti.parse_poses(str(f))
    np.testing.assert_allclose(Q, P, rtol=1e-9)
    assert kitti.parse_poses(str(f), stop=3).shape == (3, 12)


def test_error_report_quirks():
    """gt translation z and gt rotation entries (2,0),(0,2) are negated; rotations are compared against gt_R[i] (not i+offset)."""
    n = 6
    gt = np.zeros((n, 12)); gt[:, [0, 5, 10]] = 1.0
    gt[:, 11] = np.arange(n) * 1.0            # forward motion along +z
    gt[:, 2] = 0.1; gt[:, 8] = -0.1           # (0,2) and (2,0)
    est = np.zeros((n, 12)); est[:, [0, 4, 8]] = 1.0
    est[:, 2] = -0.1; est[:, 6] = 0.1         # est R (row-major 9) = gt R with the two entries negated
    est[:, 11] = -np.arange(n) * 1.0          # the pipeline's forward axis is -z (Q14)
    r = kitti.error_report(est, gt, init_offset=0)
    assert abs(r["t total"]) < 1e-12 and abs(r["R total"]) < 1e-12
    r1 = kitti.error_report(est[:-1], gt, init_offset=1)   # offset: translations against gt[i+1] -> 1 m each
    np.testing.assert_allclose(r1["t total"], n - 2)


def test_load_sequence_from_a_synthetic_kitti_tree(tmp_path):
    from PIL import Image
    frames, gt = pmv.synth_sequence(1003, 0, 3, 160, 96, 120.0, 120.0, 80.0, 48.0, nthreads=2)
    root = tmp_path
    os.makedirs(root / "sequences" / "07" / "image_0"); os.makedirs(root / "poses")
    for i, fr in enumerate(frames):
        Image.fromarray(fr).save(root / "sequences" / "07" / "image_0" / ("%06d.png" % i))
    (root / "sequences" / "07" / "calib.txt").write_text("P0: 120 0 80 0 0 120 48 0 0 0 1 0\n")
    kitti.write_poses_kitti(str(root / "poses" / "07.txt"), np.hstack([gt[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]], gt[:, [3, 7, 11]]]))
    f2, p2, K = kitti.load_sequence(str(root), "07")
    assert np.array_equal(f2, frames)
    np.testing.assert_allclose(p2, gt, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(K, [[120, 0, 80], [0, 120, 48], [0, 0, 1]])
