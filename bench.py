#!/usr/bin/env python3
"""VO frames/s of the MI355X hot path.

  python bench.py --gpus N --steps K --warmup W [--config 1|2|3|5] [--subseq L]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Default workload = BASELINE.json configs[1], the configuration the metric is quoted on: synthetic KITTI-like 1241x376 mono
sequence, 1101 frames, 400 tracked features (tol 150), bundle_size 5, 5 LM iterations. --config 2 / 3 = configs[2] / configs[3]
(4541 frames, 800 tracks, bundle 10 / 1920x1080, 1000 frames, 2000 tracks, bundle 20); --config 5 = configs[4]: eight sequences
with the KITTI 00-07 lengths, cut into independent subsequences of --subseq frames (0 = whole sequences), dealt longest-first
to the ranks by sharding.cut / assign_pieces and run B-per-GPU through the batch engine (strong scaling).

A "step" is one pass of the whole pipeline over the rank's sequence(s): pyramid build for every frame, per-frame LK /
re-detection, lag-2 PnP | triangulation, BA, poses back on the host. `value` (SURVEY.md §8d timed region) times K steps that start
from gray frames in HOST memory (streamed ingest, ingest.hip) - barrier + device sync on both sides, max over ranks; the same K
steps from HBM-resident frames are reported as `hbm_resident`. Sequences are independent: with N GPUs every rank runs its own
(weak scaling, no data-path collective); the only collective is the RCCL all-gather of the pose arrays. Every rank also runs the
`batched` leg (B distinct sequences per GPU through batched launches); rank 0 prints ONE JSON line with the job totals.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K00 = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
CONFIGS = {
    # BASELINE.json configs[1] (metric config). Real KITTI seq 07 frames are 1226x370; the metric quotes 1241x376.
    1: dict(K00, name="BASELINE configs[1]", n_frames=1101, min_tracked=400, tol=150, bundle_size=5, seed=1007),
    2: dict(K00, name="BASELINE configs[2]", n_frames=4541, min_tracked=800, tol=300, bundle_size=10, seed=1000),
    3: dict(w=1920, h=1080, fx=1000.0, fy=1000.0, cx=960.0, cy=540.0, name="BASELINE configs[3]", n_frames=1000, min_tracked=2000, tol=750,
            bundle_size=20, seed=1010),
    5: dict(K00, name="BASELINE configs[4]", n_frames=None, min_tracked=400, tol=150, bundle_size=5, seed=1000),
}
INIT_FRAMES, BA_ITERATIONS = 5, 5
# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s; FP64 vector = FP64 matrix = 78.6 TFLOP/s; FP32 vector 157.3 TFLOP/s; the integer
# VALU rate is taken as the FP32 lane rate without FMA doubling (256 CU x 128 lanes x 2.4 GHz = 78.6 Top/s)
PEAK_HBM_GBS, PEAK_FP64_MFMA_TFLOPS, PEAK_FP64_VALU_TFLOPS, PEAK_FP32_VALU_TFLOPS, PEAK_VALU_TOPS = 8000.0, 78.6, 78.6, 157.3, 78.6
PNP_ITERS, EPNP_FLOP = 100, 15e3   # §8d: F_pnp = 40 * iters * M + iters * EPnP(~15 kflop)


def pyramid_levels(w, h):
    lv = []
    for _ in range(5):
        lv.append((w, h))
        w, h = (w + 1) // 2, (h + 1) // 2
        if w <= 32 or h <= 32:
            break
    return lv


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def stage_roofline_seconds(cfg, w, h, st, lk_work, n_seq_frames):
    """per stage, ALGORITHMIC bytes or flops of one pass over a sequence divided by the peak that bounds the stage
    (the larger of its HBM and its arithmetic time). st = pipeline statistics of the pass, lk_work = measured (LK iterations,
    level passes, tracks) or None, n_seq_frames = images whose pyramid is built. Returns {stage: seconds}."""
    levels = pyramid_levels(w, h)
    wh = float(w * h)
    pyr_px = float(sum(a * b for a, b in levels))
    out = {}
    # B_pyr = W*H * (1 [write L0] + rho_{0..L-1} [reads] + (rho - 1) [writes]) per frame
    b_pyr = (wh + sum(a * b for a, b in levels[:-1]) + sum(a * b for a, b in levels[1:])) * n_seq_frames
    out["pyramid"] = b_pyr / (PEAK_HBM_GBS * 1e9)
    b_lk = 2.0 * pyr_px * st["lk_calls"] + 13.0 * st["lk_points"]
    if lk_work and lk_work[2]:
        ops_lk = 1024.0 * (40.0 * lk_work[1] + 14.0 * lk_work[0])
    else:   # no in-kernel count for this pass: §8d's form with 6 iterations per level pass on every level
        ops_lk = 1024.0 * (40.0 + 14.0 * 6.0) * len(levels) * st["lk_points"]
    out["lk"] = max(b_lk / (PEAK_HBM_GBS * 1e9), ops_lk / (PEAK_VALU_TOPS * 1e12))
    corners = float(cfg["min_tracked"])
    out["detect"] = st["detect_calls"] * max((wh + 8.0 * corners) / (PEAK_HBM_GBS * 1e9), 60.0 * wh / (PEAK_FP32_VALU_TFLOPS * 1e12))
    out["pnp"] = (40.0 * PNP_ITERS * st["pnp_points"] + st["pnp_calls"] * PNP_ITERS * EPNP_FLOP) / (PEAK_FP64_VALU_TFLOPS * 1e12)
    if st["ba_calls"]:
        nc = min(cfg["bundle_size"], 32)
        obs, pts = st["ba_obs"] / st["ba_calls"], st["ba_points"] / st["ba_calls"]
        per_iter = 250.0 * obs + 216.0 * obs + 2.0 * (6 * nc) ** 2 * 3.0 * pts + (6 * nc) ** 3 / 3.0
        out["ba"] = st["ba_calls"] * BA_ITERATIONS * per_iter / (PEAK_FP64_MFMA_TFLOPS * 1e12)
    else:
        out["ba"] = 0.0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs index + 1 (default 1 = metric config)")
    ap.add_argument("--subseq", type=int, default=256, help="--config 5: cut the sequences into independent subsequences of this many frames (0 = whole sequences)")
    ap.add_argument("--frames", type=int, default=0, help="override the frames per sequence (diagnostic; the line then names the reduced workload)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--sequential", action="store_true", help="one host thread instead of front-end/back-end threads")
    ap.add_argument("--tri-threads", type=int, default=8, help="host threads evaluating the triangulator's five-point RANSAC hypotheses")
    ap.add_argument("--kitti-seq", default="07", help="sequence used when KITTI_ROOT points at a KITTI odometry tree (default: synthetic data)")
    ap.add_argument("--poses-out", default="", help="write the estimated poses of the last step in KITTI format")
    ap.add_argument("--batch", type=int, default=192, help="batched leg: B independent sequences through batched launches on each GPU (0 = skip); "
                                                           "192 sequences of the metric config are 234 GB of frame slots (sized for 288 GB of HBM)")
    ap.add_argument("--batch-ba-mode", type=int, default=0, choices=(0, 1), help="LM implementation of the batched leg (pmv_set_ba_mode): 0 = the launch chain the "
                    "single-sequence legs use (default; measured faster: 56.2k vs 53.1k frames/s at B=192), 1 = one workgroup per solve, ONE launch per round")
    ap.add_argument("--batch-distinct", type=int, default=16, help="distinct sequences (4 seeds x start offsets) cycled over the B slots of the batched leg")
    ap.add_argument("--batch-contexts", type=int, default=0, help="diagnostic: also run the round-1 form (B contexts x 2 host threads x 2 streams)")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the hbm_resident leg (frames already staged in HBM)")
    ap.add_argument("--no-cpu-multi", action="store_true", help="skip the same-cores multi-sequence CPU baseline of the batched leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # one rank per GPU: N > 1 must be started through torch.distributed.run (nothing is re-exec'ed from here)
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`\n")
        sys.exit(2)
    dist = None
    torch = None
    tdev = "cpu"
    if world > 1:
        # torch first: it brings its own libamdhip64 and the product library must bind to the same HIP runtime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal aid (one-GPU box): PMV_BENCH_BACKEND=gloo runs the N-rank flow with CPU tensors, ranks share the visible GPUs
        backend = os.environ.get("PMV_BENCH_BACKEND", "nccl")
        dev_index = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
            tdev = "cuda"
        else:
            dist.init_process_group(backend=backend)
        local_rank = dev_index

    # The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise; the batch
    # engine uses 5 streams + the context's 3.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    # A one-GPU box share is 16 CPUs, enforced as a CFS quota over all 256 hardware threads: unpinned, ~70 runnable threads spread
    # over the machine, burn the quota in bursts and run cache-cold (measured: 672 us of CPU per frame in the batched leg, 457 us
    # pinned). Every rank therefore pins itself to its own 16 CPUs — the CPU baselines run under the same mask.
    pin_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not os.environ.get("PMV_BENCH_NO_PIN"):
        try:
            allowed = sorted(os.sched_getaffinity(0))
            if len(allowed) >= 16 * (pin_rank + 1) and len(allowed) > 16:
                os.sched_setaffinity(0, allowed[16 * pin_rank: 16 * (pin_rank + 1)])
        except (AttributeError, OSError):
            pass
    pmv = importlib.import_module("practical-multi-view_amd")
    sh = importlib.import_module("practical-multi-view_amd.sharding")
    if not os.path.exists(pmv.lib_path()):
        importlib.import_module("practical-multi-view_amd.build").build_all()
    import numpy as np

    def allreduce(x, op):
        """scalar all-reduce over the ranks (identity when N = 1)"""
        if dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return float(t.item())

    cfg = dict(CONFIGS[args.config])
    w, h = cfg["w"], cfg["h"]
    try:
        ncpu = max(1, min(16, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        ncpu = max(1, min(16, os.cpu_count() or 1))
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    # ---- the rank's sequences ------------------------------------------------------------------------------------------------
    t0 = time.time()
    data_kind = "synthetic"
    pieces_all = None
    if args.config == 5:
        lengths = [args.frames or L for L in sh.KITTI_LENGTHS]
        pieces_all = sh.cut(lengths, args.subseq, min_len=INIT_FRAMES + 3)
        mine = sh.assign_pieces(pieces_all, world)[rank]
        seqs = [(1000 + sid, start, n) for sid, start, n in mine]          # (seed, first frame, frames)
        scaling = "strong"
    else:
        seqs = [(cfg["seed"] + rank, 0, args.frames or cfg["n_frames"])]   # every rank gets its own sequence of the same shape
        scaling = "weak"
    data = []
    kroot = os.environ.get("KITTI_ROOT", "")
    if args.config == 1 and kroot and os.path.isdir(os.path.join(kroot, "sequences", args.kitti_seq)):
        # real data: same config on KITTI odometry images, calibration and ground truth from the dataset files
        kitti = importlib.import_module("practical-multi-view_amd.kitti")
        frames, gt, Km = kitti.load_sequence(kroot, args.kitti_seq, seqs[0][2])
        _, h, w = frames.shape
        K = Km.reshape(9).copy()
        data.append((frames, gt))
        data_kind = f"KITTI odometry sequence {args.kitti_seq} ({frames.shape[0]} frames, {w}x{h}) from KITTI_ROOT"
    else:
        for seed, first_frame, n in seqs:
            data.append(pmv.synth_sequence(seed, first_frame, n, w, h, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=ncpu))
    t_gen = time.time() - t0
    n_max = max(f.shape[0] for f, _ in data)
    ba_cams = max(32, cfg["bundle_size"] + 2)
    big = cfg["min_tracked"] > 1000   # configs[3]: 2000 tracks x bundle 20 -> up to ~9 k window landmarks, ~55 k observations
    ctx_kw = dict(n_slots=n_max, max_tracks=8192 if big else 4096, max_ba_cams=ba_cams, max_ba_points=32768 if big else 8192,
                  max_ba_obs=262144 if big else 65536, device=local_rank)
    # One sequence per rank: one context. Several (sub)sequences on a rank (configs[4]): ONE context holds them back to back and the
    # step runs them together through the batch engine (at most 128 at a time).
    multi = len(data) > 1 or args.config == 5
    seq_slots, first = [], 0
    if multi:
        ctx_kw = dict(ctx_kw, max_tracks=1024, max_ba_cams=8, max_ba_points=4096, max_ba_obs=32768)
    ctx = pmv.Context(w, h, **dict(ctx_kw, n_slots=sum(f.shape[0] for f, _ in data)))
    for frames, gt_ in data:
        ctx.frames_stage(first, frames)
        seq_slots.append((first, frames.shape[0], gt_))
        first += frames.shape[0]
    ctxs = [ctx]
    tri_threads = 1 if args.sequential else max(1, min(args.tri_threads, ncpu - 2))
    run_kw = dict(min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES, bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS,
                  threaded=0 if args.sequential else 1, want_features=False, n_threads=tri_threads, defer_free=True)
    batch_kw = dict(min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES, bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS,
                    want_features=False, defer_free=True)
    WAVE = 128   # (sub)sequences in flight per pmv_pipeline_run_batch call

    def step(host=False):
        """one pass over the rank's sequence(s); host=True: frames streamed from host memory (single sequence only)"""
        if multi:
            out = []
            for i in range(0, len(seq_slots), WAVE):
                out += ctx.pipeline_run_batch(seq_slots[i: i + WAVE], w, h, K, threaded=0 if args.sequential else 1, **batch_kw)
            return out
        frames, gt = data[0]
        return [ctx.pipeline_run(frames.shape[0], w, h, K, gt, build_pyramids=1, host_frames=frames if host else None, **run_kw)]

    def sync_all():
        for c in ctxs:
            c.sync()
        if dist is not None:
            if tdev == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    def gather(results):
        """final pose concatenation of the job: one RCCL all-gather over xGMI of the (padded) pose arrays when N > 1"""
        if dist is None:
            return None
        mine_arr = np.concatenate([r.poses for r in results]) if results else np.zeros((0, 12))
        return sh.gather_poses(dist, mine_arr, max_frames=sum(sh.KITTI_LENGTHS) if args.config == 5 else n_max + 8,
                               device=torch.device(tdev) if tdev == "cuda" else None)

    def timed(host):
        sync_all()
        t_start = time.perf_counter()
        kept = []   # native results are freed after the timed region (host-container teardown is not part of the path)
        for _ in range(args.steps):
            res = step(host)
            kept.append(res)
            gather(res)
        sync_all()
        el = time.perf_counter() - t_start
        # frames that went through addFrame + estimatePose per step (every step does the same work), summed over the ranks
        total = args.steps * sum(f.shape[0] - int(r_.stats["init_offset"]) for (f, _), r_ in zip(data, kept[-1]))
        el = allreduce(el, "max")
        total = int(round(allreduce(total, "sum")))
        for rs in kept[:-1]:
            for r_ in rs:
                r_.free()
        return el, total, kept[-1]

    # ---- warm-up; the per-kernel HIP-event table is taken in the LAST warm-up step (two events per launch cost host time and a
    # queue barrier, ~7000 per step), the timed region records only the critical-path kernel whose live average `roofline` reports
    prof_warm, lk_work = None, None
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last:
            ctx.prof_enable(True)
            ctx.lk_counters(reset=True)
        rs = step()
        if last:
            ctx.sync()
            ctx.prof_enable(False)
            prof_warm = ctx.prof_read()
            lk_work = ctx.lk_counters()
        for r_ in rs:
            r_.free()
    # second diagnostic pass: the kernels INSIDE the LM launch chain, one by one (their events lengthen the chain, so separately)
    prof_chain = None
    if args.warmup > 0 and not multi:
        ctx.prof_enable(True)
        ctx.prof_select(["k_bam_eval0", "k_bam_campoint", "k_bam_gemm", "k_bam_solve", "k_bam_backsub", "k_bam_finish"])
        rs_ = step()
        ctx.sync()
        ctx.prof_enable(False)
        prof_chain = ctx.prof_read()
        for r_ in rs_:
            r_.free()

    # critical path of one sequence = the back-end's serial chain (PnP(k+1) needs BA(k)'s landmarks; the front-end overlaps it):
    # the single kernel with the most time ON THAT CHAIN is the one reported
    backend_kernels = ("k_pnp_hyp", "k_pnp_select_refit", "k_tri_dlt", "k_bam_solve", "k_bam_campoint", "k_bam_gemm", "k_bam_backsub", "k_bam_eval0",
                       "k_pnp_gather", "k_ba_gather")
    cand = {}
    for src in (prof_warm or {}, prof_chain or {}):
        for k_, v in src.items():
            if k_ in backend_kernels:
                cand[k_] = v
    dom = max(cand.items(), key=lambda kv: kv[1][1])[0] if cand else None
    ctx.prof_enable(True)
    if dom:
        ctx.prof_select([dom] if not dom.startswith("k_bam_") else ["k_pnp_hyp"])   # never lengthen the timed chain with in-chain events
        dom_timed = dom if not dom.startswith("k_bam_") else "k_pnp_hyp"
    else:
        dom_timed = None
    # headline: §8d's timed region - gray frames in host memory -> poses on the host (one sequence: streamed ingest; several
    # (sub)sequences per rank: frames staged beforehand, the line says so)
    elapsed, frames_total, last_res = timed(host=not multi)
    ctx.prof_enable(False)
    prof_timed = ctx.prof_read()
    res = last_res[0]
    hbm_leg = None
    if not args.no_hbm_leg and not multi:
        el_h, tot_h, last_h = timed(host=False)
        if rank == 0:
            same = all(np.array_equal(a.poses, b.poses) for a, b in zip(last_h, last_res))
            hbm_leg = dict(value=round(tot_h / el_h, 3), unit="frames/s", ms_per_step=round(el_h / args.steps * 1e3, 3),
                           identical_to_headline_run=bool(same),
                           how="the same K steps with the gray frames already staged in HBM (pmv_frames_stage before the timed region)")
        for r_ in last_h:
            r_.free()
    for c in ctxs:
        c.pipeline_drain()

    value = frames_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_step = frames_total // args.steps
    st = res.stats
    st_sum = {k_: sum(float(r_.stats[k_]) for r_ in last_res) for k_ in st}   # over the rank's (sub)sequences
    n0 = data[0][0].shape[0]

    # ---- batched leg, on EVERY rank: B independent, DISTINCT sequences per GPU through batched launches -----------
    batched = None
    bat_local = None
    if args.batch > 1 and args.config == 1 and not kroot:
        B = args.batch
        n = args.frames or cfg["n_frames"]
        D = max(1, min(args.batch_distinct, B))
        n_seed = max(1, (D + 3) // 4)
        OFF = 40                                      # start offsets 0, 40, 80, 120 frames into each generated sequence
        tg = time.time()
        gen = []
        for k_ in range(n_seed):
            per = min(4, D - 4 * k_)
            gen.append(pmv.synth_sequence(cfg["seed"] + 64 + 4 * rank + k_, 0, n + OFF * (per - 1), w, h, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=ncpu))
        distinct = [(gen[d // 4][0][OFF * (d % 4): OFF * (d % 4) + n], gen[d // 4][1][OFF * (d % 4): OFF * (d % 4) + n]) for d in range(D)]
        t_gen_b = time.time() - tg
        # every distinct sequence's own single run (the check of the batched results): through the rank's one-sequence context
        single = []
        ctx.set_ba_mode(args.batch_ba_mode)   # the single runs the batched results are compared with use the batched leg's LM implementation
        for fr_d, gt_d in distinct:
            ctx.frames_stage(0, fr_d)
            r1 = ctx.pipeline_run(n, w, h, K, gt_d, build_pyramids=1, **dict(run_kw, defer_free=False))
            single.append((r1.poses.copy(), dict(r1.stats)))
        ctx.set_ba_mode(0)
        ctx.frames_stage(0, data[0][0])
        bc = None
        while bc is None:
            try:
                bc = pmv.Context(w, h, **dict(ctx_kw, n_slots=B * n, max_tracks=1024, max_ba_cams=8, max_ba_points=4096, max_ba_obs=32768))
            except pmv.PmvError:      # not enough free HBM for B sequences: fewer
                if B <= 8:
                    raise
                B = B - 32 if B > 64 else B // 2
        bc.set_ba_mode(args.batch_ba_mode)
        for b in range(B):
            bc.frames_stage(b * n, distinct[b % D][0])   # every slot range holds its own copy (B sequences = B x 1.7 GB of HBM, as real data would)
        bseqs = [(b * n, n, distinct[b % D][1]) for b in range(B)]
        bthr = 0 if args.sequential else 1   # per sequence: the reference's front-end / back-end threads (two requests in flight) or one thread
        warm = bc.pipeline_run_batch(bseqs, w, h, K, threaded=bthr, **batch_kw)
        bc.sync()
        s0 = bc.batch_stats()
        bc.lk_counters(reset=True)
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        c1 = time.process_time()
        results = bc.pipeline_run_batch(bseqs, w, h, K, threaded=bthr, **batch_kw)
        bc.sync()
        dtb = time.perf_counter() - t1
        cpu_b = time.process_time() - c1
        s1 = bc.batch_stats()
        lk_work_b = bc.lk_counters()
        same = all(np.array_equal(results[b].poses, single[b % D][0]) for b in range(B))
        fr_b = sum(n - int(r_.stats["init_offset"]) for r_ in results)
        bat_local = dict(frames=fr_b, seconds=dtb, cpu=cpu_b, same=same, B=B)
        bprof, rounds = {}, {}
        if rank == 0:
            # kernel table of the batched launches: a third pass with per-launch events on (they cost host time, so not in the timed pass)
            bc.prof_enable(True)
            prof_pass = bc.pipeline_run_batch(bseqs, w, h, K, threaded=bthr, **batch_kw)
            bc.sync()
            bc.prof_enable(False)
            bprof = bc.prof_read()
            for r_ in prof_pass:
                r_.free()
            for role in s1:
                d_ = {k_: s1[role][k_] - s0[role][k_] for k_ in s1[role]}
                if d_["launches"]:
                    rounds[role] = dict(launch_rounds=d_["launches"], requests_per_round=round(d_["requests"] / d_["launches"], 2),
                                        avg_round_us=round(d_["work_s"] / d_["launches"] * 1e6, 1), gpu_wait_s=round(d_["sync_s"], 3))
        tot_frames = allreduce(fr_b, "sum")
        max_sec = allreduce(dtb, "max")
        all_same = allreduce(0.0 if same else 1.0, "sum") == 0.0
        if rank == 0:
            bk = {k_: dict(launches=v[0], avg_us=round(v[1] / v[0] * 1e3, 2)) for k_, v in bprof.items()}
            # §8d frame-rate roofline of the batched leg: the stage sums of all B sequences over the measured time of the pass
            st_b = {k_: sum(float(r_.stats[k_]) for r_ in results) for k_ in ("lk_calls", "lk_points", "detect_calls", "pnp_calls", "pnp_points", "ba_calls", "ba_obs", "ba_points")}
            stage_b = stage_roofline_seconds(cfg, w, h, st_b, lk_work_b, float(B * n))
            lk_round = rounds.get("lk", {})
            b_roof = None
            if "k_lk" in bprof and lk_round and lk_work_b[2]:
                t_l = bprof["k_lk"][1] / bprof["k_lk"][0] * 1e-3
                ops = 1024.0 * (40.0 * lk_work_b[1] + 14.0 * lk_work_b[0]) / max(1, lk_round["launch_rounds"])
                b_roof = dict(kernel="k_lk_batch", bound="valu-int", achieved=round(ops / t_l / 1e12, 4), peak=PEAK_VALU_TOPS, unit="Top/s", frac=ops / t_l / 1e12 / PEAK_VALU_TOPS,
                              avg_launch_us=round(t_l * 1e6, 1), algorithmic_per_launch=round(ops, 1), traffic=None,
                              note=f"OPS_lk (measured iterations) of the {lk_round['requests_per_round']} sequences' tracks per launch on average")
            batched = dict(sequences=B, distinct_sequences=D, value=round(tot_frames / max_sec, 3), unit="frames/s", seconds=round(max_sec, 3),
                           n_gpus=world, per_gpu=round(tot_frames / max_sec / world, 3), identical_to_single_run=bool(all_same),
                           ba_mode=args.batch_ba_mode,
                           ba_mode_note="pmv_set_ba_mode: 0 = the 23-launch LM chain, problems of a round side by side in each launch; 1 = the whole solve in one "
                                        "workgroup per problem, one launch per round (slower here: profiles/r03_batch_exp_n.log); the single runs of the "
                                        "bitwise check use the same mode",
                           distinct_how=f"{n_seed} seeds x start offsets 0/{OFF}/{2 * OFF}/{3 * OFF} frames, cycled over the {B} slot ranges; every batched result "
                                        f"is compared bitwise with the single-sequence run of the same input",
                           host_threads=(2 if bthr else 1) * B + 6, host_cpu_us_per_frame=round(cpu_b / fr_b * 1e6, 1), host_cores_busy=round(cpu_b / dtb, 2),
                           frame_rate_roofline=dict(frac=sum(stage_b.values()) / dtb, stage_seconds={k_: round(v, 6) for k_, v in stage_b.items()},
                                                    measured_seconds=round(dtb, 4), note="rank 0's pass: sum over its B sequences of stage(algorithmic / peak) / measured time"),
                           combiners=rounds, kernels=bk, roofline=b_roof, input_generation_s=round(t_gen_b, 2),
                           how="pmv_pipeline_run_batch on every rank: front-end + back-end host thread per sequence (unchanged adapters), combiner threads merge the plugin "
                               "calls into batched launches (k_lk_batch, detectors, k_pnp_*_batch, k_bamB_* chain, k_tri_dlt_batch), one HIP stream per class; "
                               "value = frames of all ranks / slowest rank's time")
        for r_ in warm + results:
            r_.free()
        bc.close()
        # ---- like-for-like CPU leg of the batched figure: min(B, 16) of the SAME distinct sequences side by side on the SAME 16 cores ----
        if rank == 0 and world == 1 and args.cpu_seconds > 0 and not args.no_cpu_multi:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc_binding as ob
            lib_fast = None
            try:
                subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "fast"], check=True, timeout=300)
                lib_fast = C.CDLL(os.path.join(ROOT, "oracle", "liborc_fast.so"))
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"bench.py: fast oracle build failed ({e}); timing the -O2 build\n")
            okw = dict(min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES, bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS,
                       fast=True, lib=lib_fast, want_features=False)
            n_cpu = int(min(n, max(60, 120.0 * args.cpu_seconds / 2.0)))   # bounded sample: the first n_cpu frames of every sequence

            def cpu_multi(conc, nthr, threaded):
                outs = [None] * conc
                start = threading.Barrier(conc + 1)

                def work(i):
                    fr_d, gt_d = distinct[i % D]
                    start.wait()
                    outs[i] = ob.run_pipeline(fr_d[:n_cpu], K, gt_d[:n_cpu], threaded=threaded, n_threads=nthr, **okw)
                th = [threading.Thread(target=work, args=(i,)) for i in range(conc)]
                for t_ in th:
                    t_.start()
                start.wait()
                ta = time.perf_counter()
                for t_ in th:
                    t_.join()
                dt_ = time.perf_counter() - ta
                fr_ = sum(n_cpu - int(o_.stats["init_offset"]) for o_ in outs)
                first_flip = []   # per sequence: frames whose poses agree with the GPU run's to 1e-6 (a RANSAC consensus flips sooner or later)
                for i in range(conc):
                    a_, b_ = outs[i].poses, single[i % D][0][: len(outs[i].poses)]
                    bad_ = np.nonzero(np.abs(a_[: len(b_)] - b_).max(axis=1) > 1e-6)[0]
                    first_flip.append(int(bad_[0]) if len(bad_) else int(len(b_)))
                return dict(sequences=conc, threads_per_sequence=nthr, schedule="front-end + back-end threads" if threaded else "one thread", value=round(fr_ / dt_, 3),
                            seconds=round(dt_, 3), poses_agree_1e6_with_gpu_until_frame_min=int(min(first_flip)))
            legs = {}
            c16 = min(B, 16, ncpu)
            legs[f"{c16}x1"] = cpu_multi(c16, 1, 0)
            if ncpu >= 8:
                legs[f"{ncpu // 4}x4"] = cpu_multi(ncpu // 4, 4, 1)
            best = max(legs, key=lambda k_: legs[k_]["value"])
            batched["cpu_baseline_same_cores"] = dict(value=legs[best]["value"], unit="frames/s", best=best, cores=ncpu, legs=legs, kind="port",
                                                      sample=f"first {n_cpu} frames of the batched leg's distinct sequences, oracle pipelines (-O3 -march=native build) side by "
                                                             f"side under the same {ncpu}-CPU affinity mask as the GPU run's host threads; the 1x{ncpu} form is `cpu_baseline`")
        del gen, distinct

    if rank != 0:
        for c in ctxs:
            c.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-stage roofline entries ----------------------------------------------------------------------------
    prof = dict(prof_warm or {})
    prof.update(prof_chain or {})
    prof.update({k_: v for k_, v in prof_timed.items()})
    kern = {k_: dict(launches=v[0], total_ms=round(v[1], 4), avg_us=round(v[1] / v[0] * 1e3, 3), max_us=round(v[2] * 1e3, 3)) for k_, v in prof.items()}
    levels = pyramid_levels(w, h)
    pyr_px = sum(a * b for a, b in levels)
    lk_n = st_sum["lk_points"] / max(st_sum["lk_calls"], 1)
    pnp_m = st_sum["pnp_points"] / max(st_sum["pnp_calls"], 1)
    ba_obs = st_sum["ba_obs"] / max(st_sum["ba_calls"], 1)
    ba_pts = st_sum["ba_points"] / max(st_sum["ba_calls"], 1)
    nc = min(cfg["bundle_size"], 32)
    traffic_db = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):   # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE)
        with open(tpath) as f:
            traffic_db = json.load(f)

    def avg_s(name):
        return prof[name][1] / prof[name][0] * 1e-3 if name in prof and prof[name][0] else None

    def entry(kernel, bound, work, peak, unit, note, extra=None):
        t = avg_s(kernel)
        if t is None:
            return None
        ach = work / t / (1e9 if unit == "GB/s" else 1e12)
        tj = traffic_db.get("kernels", {}).get(kernel, {}) if traffic_db.get("config") == args.config else {}
        e = dict(kernel=kernel, bound=bound, achieved=round(ach, 5), peak=peak, unit=unit, frac=ach / peak, avg_launch_us=round(t * 1e6, 3),
                 algorithmic_per_launch=round(work, 1), traffic=tj.get("hbm_bytes_per_launch"), note=note)
        if extra:
            e.update(extra)
        return e

    stages = []
    # pyramid (HBM streaming stencil): 2.64*W*H per frame = read L0..L-1 + write padded L0..L; all n frames per launch set
    pad_bytes = sum((a + 128) * (b + 128) for a, b in levels)
    # (§8d counts the gray image once: write L0 + read L0..L-1 + write L1..L = 2.64 W*H; the kernels also read the tight gray copy)
    pyr_alg = (w * h + sum(a * b for a, b in levels[:-1]) + sum(a * b for a, b in levels[1:])) * float(n0)
    t_pyr = sum(prof[k_][1] for k_ in ("k_pad_level0", "k_pyrdown") if k_ in prof) * 1e-3 / max(1, prof.get("k_pad_level0", (1,))[0])
    if t_pyr > 0 and not multi:
        stages.append(dict(kernel="k_pad_level0+k_pyrdown", bound="hbm", achieved=round(pyr_alg / t_pyr / 1e9, 3), peak=PEAK_HBM_GBS, unit="GB/s",
                           frac=pyr_alg / t_pyr / 1e9 / PEAK_HBM_GBS, avg_launch_us=round(t_pyr * 1e6, 1), algorithmic_per_launch=round(pyr_alg, 1),
                           traffic=None, note=f"all {n0} frames per launch set; writes incl. the 64-px padding are {pad_bytes * n0} B"))
    # k_lk: HBM view (both pyramids once + 13 B per track) and the VALU-integer view with MEASURED iteration counts (OPS_lk, §8d)
    stages.append(entry("k_lk", "hbm", 2.0 * pyr_px + 13.0 * lk_n, PEAK_HBM_GBS, "GB/s",
                        "latency-bound: one frame's tracks depend on the previous frame's result, ~300 blocks in flight; a launch ends with its slowest track"))
    if lk_work and lk_work[2]:
        it, lev, trk = lk_work
        ops_per_launch = 1024.0 * (40.0 * lev + 14.0 * it) / max(1, prof.get("k_lk", (1,))[0])
        e = entry("k_lk", "valu-int", ops_per_launch, PEAK_VALU_TOPS, "Top/s", "OPS_lk = sum over tracks and levels of 1024*(40 + 14*iterations), iterations measured in-kernel",
                  dict(iterations_per_level_pass=round(it / max(lev, 1), 2), level_passes_per_track=round(lev / max(trk, 1), 2)))
        if e:
            stages.append(e)
    # detectors: §8d's B_det = W*H read + 8 B per corner written, per launch over the whole grid (what a fused pass moves)
    b_det = float(w * h) + 8.0 * cfg["min_tracked"]
    for k_ in ("k_gftt_cand", "k_gftt_pick"):
        stages.append(entry(k_, "hbm", b_det, PEAK_HBM_GBS, "GB/s", "B_det = W*H + 8 B/corner for the whole detector pass; every kernel of the pass is booked against it"))
    # PnP: FP64 VALU work of §8d, F_pnp = 40*iters*M + iters*EPnP(15 kflop), all of it in k_pnp_hyp (one wavefront per hypothesis)
    f_pnp = 40.0 * PNP_ITERS * pnp_m + PNP_ITERS * EPNP_FLOP
    stages.append(entry("k_pnp_hyp", "valu-f64", f_pnp, PEAK_FP64_VALU_TFLOPS, "TFLOP/s",
                        f"latency: serial FP64 algebra of 5-point EPnP (12x12 Jacobi, pseudo-inverses, Gauss-Newton), one wavefront per hypothesis; "
                        f"F_pnp = 40*{PNP_ITERS}*M + {PNP_ITERS}*15e3 flop with M = {pnp_m:.0f}"))
    stages.append(entry("k_pnp_select_refit", "valu-f64", 20 * 2.0 * 12 * 12 * pnp_m, PEAK_FP64_VALU_TFLOPS, "TFLOP/s",
                        "latency: sequential LM passes of the refit (<= 20 x J^T J over the inliers, 2*12*12 flop per point and pass)"))
    stages.append(entry("k_tri_dlt", "hbm", (33.0 + 132.0) * 1.5 * cfg["min_tracked"], PEAK_HBM_GBS, "GB/s", "one thread per (candidate, correspondence)"))
    # BA: FP64 MFMA flops of the two matrix kernels; F_schur = 2*(6 nc)^2 * 3P (§8d), camera blocks U_c = F^T F: 2 * 8^2 * 2*obs (padded 6+rhs -> 8)
    stages.append(entry("k_bam_gemm", "mfma", 2.0 * (6 * nc) * (6 * nc + 1) * 3.0 * ba_pts, PEAK_FP64_MFMA_TFLOPS, "TFLOP/s",
                        f"Schur contraction S -= Y W^T on v_mfma_f64_16x16x4_f64: 2*(6nc)(6nc+1)*3P with nc={nc}, P={ba_pts:.0f} (avg)"))
    stages.append(entry("k_bam_campoint", "mfma", 2.0 * 7 * 7 * 2.0 * ba_obs, PEAK_FP64_MFMA_TFLOPS, "TFLOP/s",
                        "camera blocks U_c | rhs_c = F^T F on FP64 MFMA + per-point E, E^-1, W, Y on VALU"))
    stages.append(entry("k_bam_solve", "hbm", 8.0 * ((6 * nc) ** 2 + 6 * nc), PEAK_HBM_GBS, "GB/s", "latency: blocked Cholesky of the reduced camera system in LDS, one workgroup"))
    stages.append(entry("k_bam_backsub", "hbm", 96.0 * ba_obs, PEAK_HBM_GBS, "GB/s", "per point back-substitution + r, J at the candidate (B_ba = 96 B per observation)"))
    stages.append(entry("ba_lm_chain", "hbm", 96.0 * ba_obs * BA_ITERATIONS, PEAK_HBM_GBS, "GB/s", "the whole LM launch chain of one solve (not a single kernel)"))
    stages = [e for e in stages if e]
    roofline = None
    if dom:
        roofline = next((dict(e) for e in stages if e["kernel"] == (dom_timed or dom) and e["bound"] in ("hbm", "mfma", "valu-f64")), None)
        if roofline:
            roofline["chosen_by"] = "largest single-kernel time on the critical path (back-end chain); candidates measured in the last warm-up step: " + \
                ", ".join(f"{k_} {cand[k_][1]:.1f} ms" for k_ in sorted(cand, key=lambda x: -cand[x][1])[:4])
            roofline["measured_in"] = "timed region (HIP events on the launching stream)" if (dom_timed or dom) in prof_timed else "last warm-up step"
            roofline["traffic_source"] = traffic_db.get("source")
    # §8d: frame-rate roofline fraction = sum over stages (algorithmic bytes or flops / peak) / measured time, per step of this rank
    stage_s = stage_roofline_seconds(cfg, w, h, st_sum, lk_work, float(sum(f.shape[0] for f, _ in data)))
    step_s = ms_per_step * 1e-3
    frame_rate_roofline = dict(single=dict(frac=sum(stage_s.values()) / step_s, stage_seconds={k_: round(v, 6) for k_, v in stage_s.items()},
                                           measured_seconds_per_step=round(step_s, 5)),
                               batched=batched["frame_rate_roofline"] if batched else None,
                               definition="SURVEY.md §8d: sum over stages of (algorithmic bytes or flops / the peak that bounds the stage: HBM 8 TB/s, integer VALU "
                                          "78.6 Top/s with measured LK iterations, FP32 VALU 157.3 TFLOP/s, FP64 78.6 TFLOP/s) / measured time of the pass")

    # ---- CPU baseline: the oracle pipeline, speed-oriented build (same results), on the GPU box's host cores -------------------
    cpu = None
    if args.cpu_seconds > 0 and world == 1 and not multi:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc_binding as ob
        frames, gt = data[0]
        nthr = args.cpu_threads or max(1, ncpu - 1)
        flags, lib = "-O2 (oracle/liborc.so)", None
        try:   # -O3 -march=native build for THIS machine; results must equal the plain build's
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "fast"], check=True, timeout=300)
            lib = C.CDLL(os.path.join(ROOT, "oracle", "liborc_fast.so"))
            flags = "g++ -O3 -march=native -ffp-contract=off -fno-fast-math (oracle/liborc_fast.so, built on this machine)"
        except Exception as e:   # noqa: BLE001
            sys.stderr.write(f"bench.py: fast oracle build failed ({e}); timing the -O2 build\n")
        kw = dict(min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES, bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS,
                  threaded=1, n_threads=nthr, fast=True, lib=lib)
        m = min(60, frames.shape[0])
        t1 = time.perf_counter()
        o = ob.run_pipeline(frames[:m], K, gt[:m], **kw)          # probe: sizes the sample to the time budget
        rate = (m - int(o.stats["init_offset"])) / max(float(o.stats["seconds"]), 1e-6)
        m = int(min(frames.shape[0], max(m, rate * args.cpu_seconds)))
        o = ob.run_pipeline(frames[:m], K, gt[:m], **kw)
        dt_wall = time.perf_counter() - t1
        dt = float(o.stats["seconds"])   # the pipeline run itself (setup, result extraction and teardown excluded, as for the GPU value)
        # the baseline computes the same thing: every 2-D feature of every frame equals the GPU run's (one extra, untimed GPU run)
        gfe = ctx.pipeline_run(m, w, h, K, gt[:m], **dict(run_kw, want_features=True, defer_free=False))
        feats_same = len(gfe.features) == len(o.features) and all(np.array_equal(a[:, :2], b[:, :2]) for a, b in zip(gfe.features, o.features))
        bad = np.nonzero(np.abs(gfe.poses - o.poses).max(axis=1) > 1e-6)[0]
        cpu = dict(value=round((m - int(o.stats["init_offset"])) / dt, 3), unit="frames/s", cores=nthr + 1, kind="port", flags=flags, cpu_model=cpu_model(),
                   host_cores_visible=os.cpu_count(),
                   sample=f"first {m} of {frames.shape[0]} frames of the same sequence; oracle pipeline (speed-oriented twins, bit-identical to the plain "
                          f"restatement): front-end + back-end threads, LK over {nthr} pooled threads, grid cells side by side, BA residuals on 4 threads, "
                          f"two-view geometry ahead of time on 2 helper threads (five-point RANSAC over {min(nthr, 8)} when computed inline)",
                   seconds=round(dt, 3), wall_seconds=round(dt_wall, 3), features_identical_to_gpu_run=bool(feats_same),
                   poses_agree_1e6_until_frame=int(bad[0]) if len(bad) else int(len(o.poses)),
                   stage_seconds={k_: round(float(o.stats[k_]), 3) for k_ in ("t_lk", "t_detect", "t_pnp", "t_tri", "t_ba")})
        if batched and "cpu_baseline_same_cores" in batched:
            batched["cpu_baseline_same_cores"]["legs"][f"1x{nthr + 1}"] = dict(sequences=1, threads_per_sequence=nthr + 1, value=cpu["value"], note="= cpu_baseline")
            legs = batched["cpu_baseline_same_cores"]["legs"]
            best = max(legs, key=lambda k_: legs[k_]["value"])
            batched["cpu_baseline_same_cores"].update(value=legs[best]["value"], best=best)

    if args.batch_contexts > 1 and world == 1 and args.config == 1:
        B = args.batch_contexts
        frames, gt = data[0]
        n = frames.shape[0]
        bctx = [ctx] + [pmv.Context(w, h, **dict(ctx_kw, n_slots=n)) for _ in range(B - 1)]
        for c in bctx[1:]:
            c.frames_stage(0, frames)
        results = [None] * B

        def worker(i):
            results[i] = bctx[i].pipeline_run(n, w, h, K, gt, build_pyramids=1, **dict(run_kw, n_threads=1))

        def run_all():
            th = [threading.Thread(target=worker, args=(i,)) for i in range(B)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        run_all()                          # warm-up
        for c in bctx:
            c.sync()
        warm_results = list(results)
        t1 = time.perf_counter()
        run_all()
        for c in bctx:
            c.sync()
        dtb = time.perf_counter() - t1
        for r_ in warm_results:
            r_.free()
        if batched is None:
            batched = {}
        batched["contexts_form"] = dict(sequences=B, value=round(B * (n - int(st["init_offset"])) / dtb, 3), unit="frames/s", seconds=round(dtb, 3),
                                        how="round-1 form: B contexts x (front-end + back-end host threads), 2 HIP streams each")
        ctx.pipeline_drain()
        for c in bctx[1:]:
            c.close()

    # trajectory sanity vs synthetic ground truth (z flipped: the pipeline's forward axis is -z, quirk Q14)
    gt0 = data[0][1]
    off = int(st["init_offset"])
    est = res.poses[:, 9:12]
    Rg0 = gt0[off].reshape(3, 4)[:, :3]
    g = (gt0[off: off + len(est), [3, 7, 11]] - gt0[off, [3, 7, 11]]) @ Rg0   # in the first frame's axes (a subsequence starts anywhere on the route)
    g = g * np.array([1, 1, -1])
    terr = np.linalg.norm(est - g, axis=1)

    what = "synthetic KITTI-like" if data_kind == "synthetic" else "KITTI " + args.kitti_seq
    if args.config == 5:
        n_pieces = len(pieces_all)
        workload = (f"{cfg['name']}: eight {what} sequences {w}x{h} with the KITTI 00-07 lengths {lengths}, 400 tracks (tol 150), bundle_size 5, "
                    f"cut into {n_pieces} independent subsequences of {args.subseq or 'all'} frames, dealt longest-first over {world} rank(s), "
                    f"{len(seqs)} on rank 0, up to {WAVE} in flight per GPU through the batch engine; frames staged in HBM before the timed region")
    else:
        workload = (f"{cfg['name']}: {what} sequence {w}x{h}, {n0} frames, {cfg['min_tracked']} tracks (tol {cfg['tol']}), bundle_size {cfg['bundle_size']}, "
                    f"{BA_ITERATIONS} LM iterations, init_frames {INIT_FRAMES}, GFTT+LK+EPnP-RANSAC+BA; one sequence per GPU, gray frames start in host memory")
    out = {
        "metric": "VO frames/sec on 1241x376 KITTI mono @400 tracks, bundle=5",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8/int32 front-end, f64 back-end", "data": data_kind,
        "config": {"workload": workload, "frames_per_step": frames_per_step, "pipeline_seconds_last_step": round(st["seconds"], 4),
                   "host_threads": 1 if args.sequential else 2 + (tri_threads - 1)},
        "roofline": roofline,
        "rooflines": stages,
        "frame_rate_roofline": frame_rate_roofline,
        "cpu_baseline": cpu,
        "batched": batched,
        "hbm_resident": hbm_leg,
        "timed_region": "SURVEY.md §8d: K x pmv_pipeline_run_streamed - decoded gray frames in pageable HOST memory -> pinned ring -> HBM on an ingest thread + "
                        "third stream, pyramids per 16-frame chunk, front-end + back-end, all poses back on the host (`hbm_resident`: the same K steps "
                        "from frames staged in HBM beforehand); freeing the native result objects (host containers, ~40 ms per run) happens after the "
                        "timed region for the GPU and is excluded from the CPU baseline too",
        "kernels": kern, "kernels_measured_in": "all classes: last warm-up step; k_bam_*: separate diagnostic pass; " + (f"{dom_timed}: timed region" if dom_timed else ""),
        "pipeline_stats": {k_: st_sum[k_] for k_ in ("lk_calls", "lk_points", "detect_calls", "pnp_calls", "pnp_points", "tri_calls", "ba_calls",
                                                     "ba_obs", "ba_points", "heuristic_motion", "n_landmarks")},
        "host_stage_seconds_per_step": {k_: round(st_sum[k_], 4) for k_ in ("t_lk", "t_detect", "t_pnp", "t_tri", "t_ba", "t_pnp_kernel", "t_ba_kernel",
                                                                               "t_tri_essential", "t_tri_pose", "tri_hypotheses", "tri_ahead")},
        "trajectory_error_m": {"mean": round(float(terr.mean()), 3), "max": round(float(terr.max()), 3),
                               "travelled": round(float(np.linalg.norm(g[-1])), 1), "of": "rank 0's first (sub)sequence"},
        "input_generation_s": round(t_gen, 2),
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 2)
    if batched and "cpu_baseline_same_cores" in batched:
        # like for like: B sequences on the GPU + its 16 host CPUs against sequences side by side on the same 16 CPUs
        out["batched_speedup_vs_cpu_same_cores"] = round(batched["per_gpu"] / batched["cpu_baseline_same_cores"]["value"], 2)
    if args.config != 1 or args.frames:
        out["note"] = "not the metric configuration: this line is a diagnostic for the named workload"
    kitti_mod = importlib.import_module("practical-multi-view_amd.kitti")
    if args.config != 5:
        out["reference_error_report"] = {k_: round(v, 4) for k_, v in kitti_mod.error_report(res.poses, gt0, off).items()}   # OdometryPipeline.cpp:267-296
    if args.poses_out:
        kitti_mod.write_poses_kitti(args.poses_out, res.poses)
    print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
