"""worker for tests/test_distributed_cpu.py: world_size-2 gloo run of the sequence sharding + pose gather, with the oracle
pipeline standing in for the GPU path (same host orchestration)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    import orc_binding as ob
    pmv = importlib.import_module("practical-multi-view_amd")
    sh = importlib.import_module("practical-multi-view_amd.sharding")
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lengths = [16, 11, 14, 9]                               # four tiny "sequences"
    subseq = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # > 0: independent subsequences of that many frames (sharding.cut)
    pieces = sh.cut(lengths, subseq, min_len=6)
    mine = sh.assign_pieces(pieces, world)[rank]
    w, h, f = 400, 150, 230.0
    K = np.array([f, 0, w / 2.0, 0, f, h / 2.0, 0, 0, 1.0])
    all_poses = []
    for sid, start, n in mine:
        frames, gt = pmv.synth_sequence(1000 + sid, start, n, w, h, f, f, w / 2.0, h / 2.0, nthreads=2)
        r = ob.run_pipeline(frames, K, gt, min_tracked=100, tol=40, init_frames=3, bundle_size=3, n_threads=2)
        tag = sid * 1000 + start                            # column 0 carries (sequence, first frame) of the piece
        all_poses.append(np.concatenate([np.full((len(r.poses), 1), tag, np.float64), r.poses[:, 1:]], 1))
    mine_arr = np.concatenate(all_poses) if all_poses else np.zeros((0, 12))
    gathered = sh.gather_poses(dist, mine_arr, max_frames=sum(lengths))
    if rank == 0:
        np.savez(sys.argv[1], **{f"rank{r}": g for r, g in enumerate(gathered)}, assign=np.array([len(x) for x in sh.assign_pieces(pieces, world)]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
