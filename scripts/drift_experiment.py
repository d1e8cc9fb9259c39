"""DESIGN.md §6 / VERDICT r02 item 9: how much of the trajectory drift on the metric sequence does the reference's truncation of the
LK result (OpenCVLucasKanadeFM.cpp:25) explain?  Runs the ORACLE pipeline (CPU) on the metric configuration twice:
as the reference (truncation) and with ORC_EXPERIMENT_LK_ROUND=1 (rounding), and prints both error tables in the form of the
reference's own report (OdometryPipeline.cpp:267-296).  python scripts/drift_experiment.py [n_frames]"""
import ctypes as C
import importlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(round_lk, n):
    code = f"""
import sys, json, ctypes as C, importlib, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import orc_binding as ob
pmv = importlib.import_module('practical-multi-view_amd'); kitti = importlib.import_module('practical-multi-view_amd.kitti')
c = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
frames, gt = pmv.synth_sequence(1007, 0, {n}, c['w'], c['h'], c['fx'], c['fy'], c['cx'], c['cy'], nthreads=8)
K = np.array([c['fx'], 0, c['cx'], 0, c['fy'], c['cy'], 0, 0, 1.0])
lib = C.CDLL({os.path.join(ROOT, 'oracle', 'liborc_fast.so')!r})
o = ob.run_pipeline(frames, K, gt, threaded=1, n_threads=8, fast=True, lib=lib, want_features=False)
off = int(o.stats['init_offset'])
est = o.poses[:, 9:12]
g = (gt[off: off + len(est), [3, 7, 11]] - gt[off, [3, 7, 11]]) * np.array([1, 1, -1])
terr = np.linalg.norm(est - g, axis=1)
yaw = np.degrees(np.arctan2(o.poses[:, 2], o.poses[:, 0]))
gyaw = np.degrees(np.arctan2(gt[off: off + len(est), 2], gt[off: off + len(est), 0]))
rep = kitti.error_report(o.poses, gt, off)
print(json.dumps(dict(mean=float(terr.mean()), max=float(terr.max()), travelled=float(np.linalg.norm(g[-1])), heuristic=int(o.stats['heuristic_motion']),
                      tri_calls=int(o.stats['tri_calls']), final_yaw_error_deg=float(yaw[-1] + gyaw[-1]), yaw_err_per_frame_deg=float((yaw[-1] + gyaw[-1]) / len(yaw)), report=rep)))
"""
    env = dict(os.environ)
    if round_lk:
        env["ORC_EXPERIMENT_LK_ROUND"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1101
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "fast"], check=True)
    for name, flag in (("reference (LK result truncated toward zero)", False), ("experiment (LK result rounded)", True)):
        print(name, json.dumps(run(flag, n)))
