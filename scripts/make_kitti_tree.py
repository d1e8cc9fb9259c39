"""Writes a KITTI-odometry-shaped directory tree from the synthetic corridor (to exercise the KITTI_ROOT path of bench.py without the
dataset): python scripts/make_kitti_tree.py <root> [frames] [seq]"""
import importlib, os, sys
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pmv = importlib.import_module("practical-multi-view_amd")
kitti = importlib.import_module("practical-multi-view_amd.kitti")
root = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 60; seq = sys.argv[3] if len(sys.argv) > 3 else "07"
w, h, fx, cx, cy = 1241, 376, 718.856, 607.1928, 185.2157
frames, gt = pmv.synth_sequence(1007, 0, n, w, h, fx, fx, cx, cy, nthreads=8)
os.makedirs(os.path.join(root, "sequences", seq, "image_0"), exist_ok=True)
os.makedirs(os.path.join(root, "poses"), exist_ok=True)
for i, fr in enumerate(frames):
    Image.fromarray(fr).save(os.path.join(root, "sequences", seq, "image_0", "%06d.png" % i))
with open(os.path.join(root, "sequences", seq, "calib.txt"), "w") as f:
    f.write("P0: %.12e 0 %.12e 0 0 %.12e %.12e 0 0 0 1 0\n" % (fx, cx, fx, cy))
kitti.write_poses_kitti(os.path.join(root, "poses", seq + ".txt"), np.hstack([gt[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]], gt[:, [3, 7, 11]]]))
print("wrote", n, "frames under", root)
