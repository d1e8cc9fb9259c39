"""KITTI odometry I/O and the reference's error report (OdometryPipeline.cpp:525-658 parsePoses /
parseCalibration, :267-296 error report). Used by bench.py when KITTI_ROOT points at a KITTI odometry tree
(<root>/sequences/<seq>/image_0/*.png, <root>/sequences/<seq>/calib.txt, <root>/poses/<seq>.txt); otherwise the bench runs
on the synthetic corridor. Pure host-side Python: nothing here is on the timed path."""
import glob
import os

import numpy as np


def parse_poses(path, stop=None):
    """(n, 12) float64, one KITTI pose line (row-major 3x4 [R|t]) per row — parsePoses :525-594 (reads at most `stop` lines;
    tokens beyond the 12th are ignored, missing ones stay 0 like the reference's uninitialised-but-overwritten Mats)."""
    out = []
    with open(path) as f:
        for line in f:
            if stop is not None and len(out) >= stop:
                break
            tok = line.split()
            row = np.zeros(12)
            for i, t in enumerate(tok[:12]):
                row[i] = float(t)
            out.append(row)
    return np.array(out, np.float64).reshape(-1, 12)


def parse_calibration(path, num_calib=0):
    """3x3 camera matrix from line `num_calib` of calib.txt ("P0: f 0 cx 0 0 f cy 0 0 0 1 0") — parseCalibration :596-658:
    the line is cut at single spaces; token k=1..3 -> row 0, 5..7 -> row 1, 9..11 -> row 2 (the 4th column and the label are
    skipped; a token is only consumed when a space follows it, so the last one is never read)."""
    K = np.zeros((3, 3))
    with open(path) as f:
        for i, line in enumerate(f):
            if i != num_calib:
                continue
            calib = line.rstrip("\n")
            k = 0
            while " " in calib:
                pos = calib.index(" ")
                tok = calib[:pos]
                calib = calib[pos + 1:]
                try:
                    v = float(tok)
                except ValueError:
                    v = 0.0   # stringstream >> double on "P0:" leaves 0
                if k in (1, 2, 3):
                    K[0, k - 1] = v
                elif k in (5, 6, 7):
                    K[1, k - 5] = v
                elif k in (9, 10, 11):
                    K[2, k - 9] = v
                k += 1
    return K


def load_sequence(root, seq="07", n=None, camera="image_0"):
    """frames (n, h, w) uint8, poses (n, 12), K (3, 3). Gray PNGs: imread(COLOR) + BGR2GRAY is the identity."""
    from PIL import Image   # only needed for real data
    files = sorted(glob.glob(os.path.join(root, "sequences", seq, camera, "*.png")))   # cv::glob order (:62)
    if n is not None:
        files = files[:n]
    if not files:
        raise FileNotFoundError(f"no images under {root}/sequences/{seq}/{camera}")
    first = np.asarray(Image.open(files[0]).convert("L"))
    frames = np.empty((len(files),) + first.shape, np.uint8)
    frames[0] = first
    for i, fn in enumerate(files[1:], 1):
        frames[i] = np.asarray(Image.open(fn).convert("L"))
    poses = parse_poses(os.path.join(root, "poses", f"{seq}.txt"), stop=len(files))
    K = parse_calibration(os.path.join(root, "sequences", seq, "calib.txt"), 0)
    return frames, poses, K


def write_poses_kitti(path, poses12):
    """poses (n, 12: R row-major then t) -> KITTI lines r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz"""
    P = np.asarray(poses12, np.float64).reshape(-1, 12)
    with open(path, "w") as f:
        for p in P:
            R, t = p[:9].reshape(3, 3), p[9:]
            f.write(" ".join("%.9e" % v for v in np.hstack([R, t[:, None]]).reshape(-1)) + "\n")


def error_report(est_poses12, gt_poses12, init_offset=0):
    """The reference's error report (:267-296): per pose i >= 1, ||t_i - gt_t[i+off] (z negated)|| and
    ||R_i - gt_R[i] (entries (2,0),(0,2) negated)|| (Frobenius) — note the rotation compares against gt_R[i], not i+off, and
    the sign flips are applied to gt_R[i+off] in place, as in the reference. Returns the totals/min/max/std it writes."""
    E = np.asarray(est_poses12, np.float64).reshape(-1, 12)
    G = np.array(gt_poses12, np.float64).reshape(-1, 12).copy()
    gR = G[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3, 3).copy()
    gt = G[:, [3, 7, 11]].copy()
    et, eR = [], []
    for i in range(1, len(E)):
        j = i + init_offset
        gt[j, 2] *= -1
        gR[j, 2, 0] *= -1
        gR[j, 0, 2] *= -1
        et.append(float(np.linalg.norm(E[i, 9:12] - gt[j])))
        eR.append(float(np.linalg.norm(E[i, :9].reshape(3, 3) - gR[i])))

    def sd(v):   # OdometryPipeline::standardDeviation :660-672 (n-1 normalisation)
        v = np.asarray(v)
        return float(np.sqrt(((v - v.mean()) ** 2).sum() / (len(v) - 1))) if len(v) > 1 else float("nan")
    return {"R total": float(np.sum(eR)), "R min": min(eR), "R max": max(eR), "R std": sd(eR),
            "t total": float(np.sum(et)), "t min": min(et), "t max": max(et), "t std": sd(et)}
