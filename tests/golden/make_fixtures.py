"""Generates the committed fixtures under tests/golden/ with the oracle in THIS repo (the reference ships no fixtures and
cannot be built here). Run from the repo root:  python tests/golden/make_fixtures.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc_binding as ob  # noqa: E402

pmv = importlib.import_module("practical-multi-view_amd")
HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = C.POINTER(C.c_int)


def container_order():
    lib = ob.load().lib
    rng = np.random.default_rng(42)
    out = {}
    for key, n in (("a", 40), ("b", 400)):
        cols = rng.integers(0, 1241, n).astype(np.int32)
        rows = rng.integers(0, 376, n).astype(np.int32)
        cols[n // 2] = cols[3]; rows[n // 2] = rows[3]        # a duplicate pixel (pointer-identity keys keep both)
        order = np.zeros(n, np.int32)
        lib.orc_host_map_order(cols.ctypes.data_as(_i32p), rows.ctypes.data_as(_i32p), n, order.ctypes.data_as(_i32p))
        out[f"cols_{key}"], out[f"rows_{key}"], out[f"order_{key}"] = cols, rows, order
    np.savez_compressed(os.path.join(HERE, "container_order.npz"), **out)


def pipeline_small():
    w, h, f, n, seed = 620, 188, 355.0, 24, 1006
    frames, gt = pmv.synth_sequence(seed, 0, n, w, h, f, f, w / 2.0, h / 2.0, nthreads=8)
    K = np.array([f, 0, w / 2.0, 0, f, h / 2.0, 0, 0, 1.0])
    r = ob.run_pipeline(frames, K, gt, min_tracked=200, tol=75, bundle_size=3, n_threads=8)
    np.savez_compressed(os.path.join(HERE, "pipeline_small.npz"), w=w, h=h, f=f, n=n, seed=seed,
                        frame0_sample=frames[0, ::37, ::41], counts=np.array([len(a) for a in r.features]),
                        features=np.concatenate([a[:, :2] for a in r.features]).astype(np.int16), poses=r.poses)


def frontend_vectors():
    """GFTT corner lists and LK outputs on two seeded 640x200 frames: what the GPU tests also compare against."""
    o = ob.load()
    w, h, f = 640, 200, 370.0
    frames, _ = pmv.synth_sequence(1007, 4, 2, w, h, f, f, 320.0, 100.0, nthreads=4)
    cells = pmv.grid_cells(w, h)
    corners = [o.gftt_cell(frames[0], c, 20) for c in cells]
    pts = np.concatenate([d + c[:2] for c, d in zip(cells, corners)]).astype(np.float32)
    xy, st, err, lv = o.lk_track(frames[0], frames[1], pts)
    np.savez_compressed(os.path.join(HERE, "frontend_640x200.npz"), w=w, h=h, f=f, seed=1007, first=4,
                        corner_counts=np.array([len(c) for c in corners]), corners=np.concatenate(corners).astype(np.int16),
                        lk_xy=xy, lk_status=st, lk_err=err, levels=lv)


if __name__ == "__main__":
    container_order()
    pipeline_small()
    frontend_vectors()
    print("fixtures written to", HERE)
