"""Multi-GPU layer: the VO hot path shards by SEQUENCE — independent (sub)sequences, one process per GPU, no
data-path collective. The only exchange is the final concatenation of the pose arrays: one all-gather (RCCL over xGMI with
the "nccl" backend on GPUs, gloo on CPU for tests) of a padded (max_frames, 12) float64 block per rank plus the per-rank
frame counts. At <= 450 KB per rank that transfer is latency-bound; no all-reduce or ring pipeline is involved."""
import numpy as np

# KITTI odometry sequences 00-07 frame counts (BASELINE configs[4]): one sequence per GPU
KITTI_LENGTHS = [4541, 1101, 4661, 801, 271, 2761, 1101, 1101]


def assign_sequences(lengths, world):
    """Longest-first greedy assignment of sequence ids to ranks (balances the longest shard, which bounds wall time).
    Returns a list of lists: ranks[r] = sequence ids processed by rank r, in processing order."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    ranks = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        ranks[r].append(i)
        load[r] += lengths[i]
    return ranks


def cut(lengths, L, min_len=8):
    """Cut every sequence into independent SUBSEQUENCES of L frames (the reference's own loop makes a subsequence
    "just a shorter run": OdometryPipeline.cpp:212-229 reads images init_offset+1 .. stop, :428-482 `initialise` picks the start
    among the first init_frames images of whatever list it is given). Piece p of sequence s covers images [start, start + n) of
    s and is a complete run of its own: own initialise, own landmarks, poses relative to its own first pose. A tail shorter
    than min_len (a run needs init_frames + 2 images) is merged into the piece before it. L <= 0: no cutting.
    Returns [(sequence id, start, n)] in (sequence, start) order."""
    pieces = []
    for sid, n in enumerate(lengths):
        if L <= 0 or n <= L:
            pieces.append((sid, 0, n))
            continue
        starts = list(range(0, n, L))
        if n - starts[-1] < min_len:
            starts.pop()
        for i, st in enumerate(starts):
            end = starts[i + 1] if i + 1 < len(starts) else n
            pieces.append((sid, st, end - st))
    return pieces


def assign_pieces(pieces, world):
    """Longest-first greedy dealing of cut() pieces to ranks: ranks[r] = pieces of rank r in processing order (longest first)."""
    a = assign_sequences([p[2] for p in pieces], world)
    return [[pieces[i] for i in idx] for idx in a]


def gather_poses(dist, poses, max_frames, device=None):
    """all-gather of variable-length pose arrays. poses: (n_i, 12) float64 numpy on every rank.
    Returns a list (one entry per rank) of (n_r, 12) numpy arrays, identical on every rank."""
    import torch
    world = dist.get_world_size()
    n = int(poses.shape[0])
    if n > max_frames:
        raise ValueError(f"{n} poses exceed max_frames={max_frames}")
    buf = torch.zeros((max_frames + 1, 12), dtype=torch.float64, device=device)
    buf[0, 0] = float(n)                                  # row 0 carries the valid-row count
    if n:
        buf[1: n + 1] = torch.from_numpy(np.ascontiguousarray(poses, np.float64)).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = []
    for t in out:
        t = t.cpu().numpy()
        k = int(round(t[0, 0]))
        res.append(t[1: k + 1].copy())
    return res
