"""N > 1 path on CPU: two gloo ranks shard four sequences (longest first), each runs its shard independently, the poses are
concatenated by one all-gather. No data-path collective exists on this hot path."""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assignment_is_longest_first_and_balanced():
    sh = importlib.import_module("practical-multi-view_amd.sharding")
    a = sh.assign_sequences(sh.KITTI_LENGTHS, 8)
    assert sorted(sum(a, [])) == list(range(8)) and all(len(x) == 1 for x in a)
    assert a[0] == [2] and a[1] == [0]                       # 4661 and 4541 frames first
    a = sh.assign_sequences(sh.KITTI_LENGTHS, 2)
    loads = [sum(sh.KITTI_LENGTHS[i] for i in x) for x in a]
    assert sorted(sum(a, [])) == list(range(8)) and abs(loads[0] - loads[1]) <= 801
    assert sh.assign_sequences([5, 5, 5], 1) == [[0, 1, 2]]


def test_two_rank_gloo_shard_and_gather(tmp_path):
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "_dist_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600)
    g = np.load(out)
    lengths = [16, 11, 14, 9]
    assert list(g["assign"]) == [2, 2]
    r0, r1 = g["rank0"], g["rank1"]
    # rank 0 got sequences {0, 3}, rank 1 {2, 1} (longest first, alternating); every sequence yields len-1-init_offset.. poses
    tags0, tags1 = set(np.unique(r0[:, 0]).astype(int)), set(np.unique(r1[:, 0]).astype(int))
    assert tags0 == {0, 3000} and tags1 == {1000, 2000}
    for sid in range(4):
        rows = (r0 if sid * 1000 in tags0 else r1)
        n = int((rows[:, 0] == sid * 1000).sum())
        assert lengths[sid] - 4 <= n <= lengths[sid] - 1
    assert r0.shape[1] == 12 and np.isfinite(r0).all() and np.isfinite(r1).all()


def test_cut_makes_independent_subsequences():
    sh = importlib.import_module("practical-multi-view_amd.sharding")
    assert sh.cut([16, 11, 14, 9], 8, min_len=6) == [(0, 0, 8), (0, 8, 8), (1, 0, 11), (2, 0, 8), (2, 8, 6), (3, 0, 9)]   # 11 = 8 + a tail of 3 < 6: merged
    assert sh.cut([16, 11], 0) == [(0, 0, 16), (1, 0, 11)] and sh.cut([5], 8) == [(0, 0, 5)]
    p = sh.cut(sh.KITTI_LENGTHS, 256, min_len=8)
    for sid, L in enumerate(sh.KITTI_LENGTHS):        # every image of every sequence is in exactly one piece, in order
        mine = [(s, n) for q, s, n in p if q == sid]
        assert mine[0][0] == 0 and sum(n for _, n in mine) == L and all(a[0] + a[1] == b[0] for a, b in zip(mine, mine[1:]))
        assert all(n >= 8 for _, n in mine)
    a = sh.assign_pieces(p, 8)
    loads = [sum(n for _, _, n in r) for r in a]
    assert sorted(sum(a, [])) == sorted(p) and max(loads) - min(loads) <= 256          # longest-first dealing balances to one piece
    assert all(r == sorted(r, key=lambda q: -q[2]) for r in a)                         # every rank starts with its longest piece


def test_two_rank_gloo_subsequences(tmp_path):
    """sharding.cut pieces over two gloo ranks: every piece is its own reference run over its image range (own initialise, poses
    relative to its own start), identical to running that range alone; the gather returns every piece exactly once."""
    out = str(tmp_path / "gathered_sub.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "tests", "_dist_worker.py"), out, "8"]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600)
    g = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc_binding as ob
    pmv = importlib.import_module("practical-multi-view_amd")
    sh = importlib.import_module("practical-multi-view_amd.sharding")
    pieces = sh.cut([16, 11, 14, 9], 8, min_len=6)
    assert list(g["assign"]) == [3, 3]
    rows = np.concatenate([g["rank0"], g["rank1"]])
    assert set(np.unique(rows[:, 0]).astype(int)) == {sid * 1000 + st for sid, st, _ in pieces}
    w, h, f = 400, 150, 230.0
    K = np.array([f, 0, w / 2.0, 0, f, h / 2.0, 0, 0, 1.0])
    for sid, st, n in pieces:
        frames, gt = pmv.synth_sequence(1000 + sid, st, n, w, h, f, f, w / 2.0, h / 2.0, nthreads=2)
        r = ob.run_pipeline(frames, K, gt, min_tracked=100, tol=40, init_frames=3, bundle_size=3, n_threads=2)
        mine = rows[rows[:, 0] == sid * 1000 + st]
        assert mine.shape[0] == len(r.poses) and np.array_equal(mine[:, 1:], r.poses[:, 1:])
        assert np.array_equal(r.poses[0], np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0]))   # each piece starts at its own identity pose
