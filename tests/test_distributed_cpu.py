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
    assert tags0 == {0, 3} and tags1 == {1, 2}
    for sid in range(4):
        rows = (r0 if sid in tags0 else r1)
        n = int((rows[:, 0] == sid).sum())
        assert lengths[sid] - 4 <= n <= lengths[sid] - 1
    assert r0.shape[1] == 12 and np.isfinite(r0).all() and np.isfinite(r1).all()
