#!/usr/bin/env python3
"""VO frames/s of the MI355X hot path.

  python bench.py --gpus N --steps K --warmup W [--config 1|2|3|5]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Default workload = BASELINE.json configs[1], the configuration the metric is quoted on: synthetic KITTI-like 1241x376 mono
sequence, 1101 frames, 400 tracked features (tol 150), bundle_size 5, 5 LM iterations. --config 2 / 3 = configs[2] / configs[3]
(4541 frames, 800 tracks, bundle 10 / 1920x1080, 1000 frames, 2000 tracks, bundle 20); --config 5 = configs[4]: eight sequences
with the KITTI 00-07 lengths dealt to the ranks by sharding.assign_sequences (strong scaling).

A "step" is one pass of the whole pipeline over the rank's sequence(s) with the gray frames already resident in HBM: pyramid
build for every frame, per-frame LK / re-detection, lag-2 PnP | triangulation, BA, poses back on the host. `value` times K steps
(barrier + device sync on both sides, max over ranks). The same steps starting from HOST memory (streamed ingest, ingest.hip)
are timed as `pcie_inclusive`. Sequences are independent: with N GPUs every rank runs its own (weak scaling, no data-path
collective); the only collective is the RCCL all-gather of the pose arrays. Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
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
PEAK_HBM_GBS, PEAK_FP64_MFMA_TFLOPS, PEAK_VALU_TOPS = 8000.0, 78.6, 78.6   # /opt/skills/guides/MI355X_MICROARCH.md; 256 CU x 128 lanes x 2.4 GHz


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="BASELINE.json configs index + 1 (default 1 = metric config)")
    ap.add_argument("--frames", type=int, default=0, help="override the frames per sequence (diagnostic; the line then names the reduced workload)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--sequential", action="store_true", help="one host thread instead of front-end/back-end threads")
    ap.add_argument("--tri-threads", type=int, default=8, help="host threads evaluating the triangulator's five-point RANSAC hypotheses")
    ap.add_argument("--kitti-seq", default="07", help="sequence used when KITTI_ROOT points at a KITTI odometry tree (default: synthetic data)")
    ap.add_argument("--poses-out", default="", help="write the estimated poses of the last step in KITTI format")
    ap.add_argument("--batch", type=int, default=128, help="extra leg: B independent sequences through batched launches on the one GPU (0 = skip); "
                                                           "128 sequences of the metric config are 222 GB of frame slots (sized for 288 GB of HBM)")
    ap.add_argument("--batch-contexts", type=int, default=0, help="diagnostic: also run the round-1 form (B contexts x 2 host threads x 2 streams)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the pcie_inclusive (streamed from host memory) leg")
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
    # pinned). Every rank therefore pins itself to its own 16 CPUs — the CPU baseline runs under the same mask.
    if not os.environ.get("PMV_BENCH_NO_PIN"):
        try:
            allowed = sorted(os.sched_getaffinity(0))
            if len(allowed) >= 16 * (local_rank + 1) and len(allowed) > 16:
                os.sched_setaffinity(0, allowed[16 * local_rank: 16 * (local_rank + 1)])
        except (AttributeError, OSError):
            pass
    pmv = importlib.import_module("practical-multi-view_amd")
    sh = importlib.import_module("practical-multi-view_amd.sharding")
    if not os.path.exists(pmv.lib_path()):
        importlib.import_module("practical-multi-view_amd.build").build_all()
    import numpy as np

    cfg = dict(CONFIGS[args.config])
    w, h = cfg["w"], cfg["h"]
    ncpu = max(1, min(16, os.cpu_count() or 1))
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    # ---- the rank's sequences ------------------------------------------------------------------------------------------------
    t0 = time.time()
    data_kind = "synthetic"
    if args.config == 5:
        lengths = [args.frames or L for L in sh.KITTI_LENGTHS]
        mine = sh.assign_sequences(lengths, world)[rank]
        seqs = [(1000 + sid, lengths[sid]) for sid in mine]
        scaling = "strong"
    else:
        seqs = [(cfg["seed"] + rank, args.frames or cfg["n_frames"])]   # every rank gets its own sequence of the same shape
        scaling = "weak"
    data = []
    kroot = os.environ.get("KITTI_ROOT", "")
    if args.config == 1 and kroot and os.path.isdir(os.path.join(kroot, "sequences", args.kitti_seq)):
        # real data: same config on KITTI odometry images, calibration and ground truth from the dataset files
        kitti = importlib.import_module("practical-multi-view_amd.kitti")
        frames, gt, Km = kitti.load_sequence(kroot, args.kitti_seq, seqs[0][1])
        _, h, w = frames.shape
        K = Km.reshape(9).copy()
        data.append((frames, gt))
        data_kind = f"KITTI odometry sequence {args.kitti_seq} ({frames.shape[0]} frames, {w}x{h}) from KITTI_ROOT"
    else:
        for seed, n in seqs:
            data.append(pmv.synth_sequence(seed, 0, n, w, h, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=ncpu))
    t_gen = time.time() - t0
    n_max = max(f.shape[0] for f, _ in data)
    ba_cams = max(32, cfg["bundle_size"] + 2)
    big = cfg["min_tracked"] > 1000   # configs[3]: 2000 tracks x bundle 20 -> up to ~9 k window landmarks, ~55 k observations
    ctx_kw = dict(n_slots=n_max, max_tracks=8192 if big else 4096, max_ba_cams=ba_cams, max_ba_points=32768 if big else 8192,
                  max_ba_obs=262144 if big else 65536, device=local_rank)
    # inputs resident in HBM before the timed region. One sequence per rank: one context. Several sequences on a rank (configs[4] on
    # fewer than 8 GPUs): ONE context holds them back to back and the step runs them together through the batch engine.
    multi = len(data) > 1
    seq_slots, first = [], 0
    ctx = pmv.Context(w, h, **dict(ctx_kw, n_slots=sum(f.shape[0] for f, _ in data)))
    for frames, gt_ in data:
        ctx.frames_stage(first, frames)
        seq_slots.append((first, frames.shape[0], gt_))
        first += frames.shape[0]
    ctxs = [ctx]
    tri_threads = 1 if args.sequential else max(1, min(args.tri_threads, ncpu - 2))
    run_kw = dict(min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES, bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS,
                  threaded=0 if args.sequential else 1, want_features=False, n_threads=tri_threads, defer_free=True)

    def step(host=False):
        """one pass over the rank's sequence(s); host=True: frames streamed from host memory (single sequence only)"""
        if multi:
            return ctx.pipeline_run_batch(seq_slots, w, h, K, min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES,
                                          bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS, want_features=False, defer_free=True,
                                          threaded=0 if args.sequential else 1)
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
        if dist is not None:
            tmax = torch.tensor([el], dtype=torch.float64, device=tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
            tsum = torch.tensor([float(total)], dtype=torch.float64, device=tdev)
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            total = int(round(float(tsum.item())))
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
    if args.warmup > 0:
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
    backend_kernels = ("k_pnp_hyp", "k_pnp_select_refit", "k_tri_dlt", "k_bam_solve", "k_bam_campoint", "k_bam_gemm", "k_bam_backsub", "k_bam_eval0")
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
    elapsed, frames_total, last_res = timed(host=False)
    ctx.prof_enable(False)
    prof_timed = ctx.prof_read()
    res = last_res[0]
    host_leg = None
    if not args.no_host_leg and not multi:
        el_h, tot_h, last_h = timed(host=True)
        if rank == 0:
            same = all(np.array_equal(a.poses, b.poses) for a, b in zip(last_h, last_res))
            host_leg = dict(value=round(tot_h / el_h, 3), unit="frames/s", ms_per_step=round(el_h / args.steps * 1e3, 3),
                            host_bytes_per_step=int(sum(f.nbytes for f, _ in data)) * world, identical_to_hbm_resident_run=bool(same),
                            how="K x pmv_pipeline_run_streamed: frames in pageable host memory -> pinned ring -> HBM on an ingest thread + third stream, "
                                "pyramids per 16-frame chunk, overlapped with tracking (SURVEY §8d timed region)")
        for r_ in last_h:
            r_.free()
    for c in ctxs:
        c.pipeline_drain()

    if rank != 0:
        for c in ctxs:
            c.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    value = frames_total / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_step = frames_total // args.steps
    st = res.stats
    n0 = data[0][0].shape[0]

    # ---- per-stage roofline entries ----------------------------------------------------------------------------
    prof = dict(prof_warm or {})
    prof.update(prof_chain or {})
    prof.update({k_: v for k_, v in prof_timed.items()})
    kern = {k_: dict(launches=v[0], total_ms=round(v[1], 4), avg_us=round(v[1] / v[0] * 1e3, 3), max_us=round(v[2] * 1e3, 3)) for k_, v in prof.items()}
    levels = pyramid_levels(w, h)
    pyr_px = sum(a * b for a, b in levels)
    lk_n = st["lk_points"] / max(st["lk_calls"], 1)
    pnp_m = st["pnp_points"] / max(st["pnp_calls"], 1)
    ba_obs = st["ba_obs"] / max(st["ba_calls"], 1)
    ba_pts = st["ba_points"] / max(st["ba_calls"], 1)
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
    pyr_alg = (w * h + sum(a * b for a, b in levels[:-1]) + sum(a * b for a, b in levels)) * float(n0)
    t_pyr = sum(prof[k_][1] for k_ in ("k_pad_level0", "k_pyrdown") if k_ in prof) * 1e-3 / max(1, prof.get("k_pad_level0", (1,))[0])
    if t_pyr > 0:
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
            e["unit"] = "Top/s"
            stages.append(e)
    stages.append(entry("k_gftt_eig", "hbm", float(w * h) * 5.0, PEAK_HBM_GBS, "GB/s", "W*H read + 4*W*H response map written"))
    stages.append(entry("k_gftt_select", "hbm", 4.0 * w * h, PEAK_HBM_GBS, "GB/s", "latency: sequential arg-max / suppression rounds per cell"))
    stages.append(entry("k_pnp_hyp", "hbm", 20.0 * pnp_m + 100 * 48.0 + 100 * pnp_m, PEAK_HBM_GBS, "GB/s",
                        "latency: serial FP64 algebra of 5-point EPnP (12x12 Jacobi, pseudo-inverses, Gauss-Newton), one wavefront per hypothesis"))
    stages.append(entry("k_pnp_select_refit", "hbm", 20.0 * pnp_m, PEAK_HBM_GBS, "GB/s", "latency: sequential LM passes of the refit"))
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
        roofline = next((dict(e) for e in stages if e["kernel"] == (dom_timed or dom) and e["bound"] in ("hbm", "mfma")), None)
        if roofline:
            roofline["chosen_by"] = f"largest single-kernel time on the critical path (back-end chain); candidates measured in the last warm-up step: " + \
                ", ".join(f"{k_} {cand[k_][1]:.1f} ms" for k_ in sorted(cand, key=lambda x: -cand[x][1])[:4])
            roofline["measured_in"] = "timed region (HIP events on the launching stream)" if (dom_timed or dom) in prof_timed else "last warm-up step"
            roofline["traffic_source"] = traffic_db.get("source")

    # ---- CPU baseline: the oracle pipeline, speed-oriented build (same results), on the GPU box's host cores -------------------
    cpu = None
    if args.cpu_seconds > 0 and world == 1:
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

    # ---- batched leg: B independent sequences on ONE GPU through batched launches (batch_engine.hip) ---------------
    batched = None
    if args.batch > 1 and world == 1 and args.config == 1:
        B = args.batch
        frames, gt = data[0]
        n = frames.shape[0]
        bc = None
        while bc is None:
            try:
                bc = pmv.Context(w, h, **dict(ctx_kw, n_slots=B * n, max_tracks=1024, max_ba_cams=8, max_ba_points=4096, max_ba_obs=32768))
            except pmv.PmvError:      # not enough free HBM for B sequences: halve
                if B <= 8:
                    raise
                B //= 2
        for b in range(B):
            bc.frames_stage(b * n, frames)     # the same frames in every slot range: a throughput leg, every run is a full independent pass
        seqs = [(b * n, n, gt) for b in range(B)]
        bkw = dict(min_tracked=cfg["min_tracked"], tol=cfg["tol"], init_frames=INIT_FRAMES, bundle_size=cfg["bundle_size"], ba_iterations=BA_ITERATIONS,
                   want_features=False, defer_free=True, threaded=0)
        warm = bc.pipeline_run_batch(seqs, w, h, K, **bkw)
        bc.sync()
        s0 = bc.batch_stats()
        t1 = time.perf_counter()
        c1 = time.process_time()
        results = bc.pipeline_run_batch(seqs, w, h, K, **bkw)
        bc.sync()
        dtb = time.perf_counter() - t1
        cpu_b = time.process_time() - c1
        s1 = bc.batch_stats()
        # kernel table of the batched launches: a third pass with per-launch events on (they cost host time, so not in the timed pass)
        bc.prof_enable(True)
        prof_pass = bc.pipeline_run_batch(seqs, w, h, K, **bkw)
        bc.sync()
        bc.prof_enable(False)
        bprof = bc.prof_read()
        same = all(np.array_equal(r_.poses, res.poses) for r_ in results)
        fr_b = sum(n - int(r_.stats["init_offset"]) for r_ in results)
        rounds = {}
        for role in s1:
            d_ = {k_: s1[role][k_] - s0[role][k_] for k_ in s1[role]}
            if d_["launches"]:
                rounds[role] = dict(launch_rounds=d_["launches"], requests_per_round=round(d_["requests"] / d_["launches"], 2),
                                    avg_round_us=round(d_["work_s"] / d_["launches"] * 1e6, 1), gpu_wait_s=round(d_["sync_s"], 3))
        bk = {k_: dict(launches=v[0], avg_us=round(v[1] / v[0] * 1e3, 2)) for k_, v in bprof.items()}
        # roofline of the batched k_lk launch: HBM view (every sequence's two pyramids once per round) and measured VALU-int work
        lk_round = rounds.get("lk", {})
        b_roof = None
        if "k_lk" in bprof and lk_round:
            t_l = bprof["k_lk"][1] / bprof["k_lk"][0] * 1e-3
            alg = lk_round["requests_per_round"] * (2.0 * pyr_px + 13.0 * lk_n)
            b_roof = dict(kernel="k_lk_batch", bound="hbm", achieved=round(alg / t_l / 1e9, 3), peak=PEAK_HBM_GBS, unit="GB/s", frac=alg / t_l / 1e9 / PEAK_HBM_GBS,
                          avg_launch_us=round(t_l * 1e6, 1), algorithmic_per_launch=round(alg, 1), traffic=None,
                          note=f"{lk_round['requests_per_round']} sequences' tracks per launch on average")
        batched = dict(sequences=B, value=round(fr_b / dtb, 3), unit="frames/s", seconds=round(dtb, 3), identical_to_single_run=bool(same),
                       host_threads=B + 5, host_cpu_us_per_frame=round(cpu_b / fr_b * 1e6, 1), host_cores_busy=round(cpu_b / dtb, 2),
                       combiners=rounds, kernels=bk, roofline=b_roof,
                       how="pmv_pipeline_run_batch: one host thread per sequence (unchanged adapters), five combiner threads merge the plugin calls "
                           "into batched launches (k_lk_batch, detectors, k_pnp_*_batch, k_bamB_* chain, k_tri_dlt_batch), one HIP stream per class")
        for r_ in warm + results + prof_pass:
            r_.free()
        bc.close()
    if args.batch_contexts > 1 and world == 1 and args.config == 1:
        import threading
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
    g = gt0[off: off + len(est), [3, 7, 11]] - gt0[off, [3, 7, 11]]
    g = g * np.array([1, 1, -1])
    terr = np.linalg.norm(est - g, axis=1)

    what = "synthetic KITTI-like" if data_kind == "synthetic" else "KITTI " + args.kitti_seq
    if args.config == 5:
        workload = (f"{cfg['name']}: eight {what} sequences {w}x{h} with the KITTI 00-07 lengths {[n for _, n in seqs] if world == 1 else sh.KITTI_LENGTHS}, "
                    f"400 tracks (tol 150), bundle_size 5, sharded longest-first over {world} rank(s)")
    else:
        workload = (f"{cfg['name']}: {what} sequence {w}x{h}, {n0} frames, {cfg['min_tracked']} tracks (tol {cfg['tol']}), bundle_size {cfg['bundle_size']}, "
                    f"{BA_ITERATIONS} LM iterations, init_frames {INIT_FRAMES}, GFTT+LK+EPnP-RANSAC+BA; one sequence per GPU")
    out = {
        "metric": "VO frames/sec on 1241x376 KITTI mono @400 tracks, bundle=5",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8/int32 front-end, f64 back-end", "data": data_kind,
        "config": {"workload": workload, "frames_per_step": frames_per_step, "pipeline_seconds_last_step": round(st["seconds"], 4),
                   "host_threads": 1 if args.sequential else 2 + (tri_threads - 1)},
        "roofline": roofline,
        "rooflines": stages,
        "cpu_baseline": cpu,
        "batched": batched,
        "pcie_inclusive": host_leg,
        "timed_region": "K x pmv_pipeline_run (pyramids of all frames rebuilt, front-end + back-end, result poses read back) from HBM-resident gray "
                        "frames; pcie_inclusive: the same K steps from frames in pageable HOST memory (streamed ingest); freeing the native result "
                        "objects (host containers, ~40 ms per run) happens after the timed region for the GPU and is excluded from the CPU baseline too",
        "kernels": kern, "kernels_measured_in": "all classes: last warm-up step; k_bam_*: separate diagnostic pass; " + (f"{dom_timed}: timed region" if dom_timed else ""),
        "pipeline_stats": {k_: st[k_] for k_ in ("lk_calls", "lk_points", "detect_calls", "pnp_calls", "pnp_points", "tri_calls", "ba_calls",
                                                 "ba_obs", "ba_points", "heuristic_motion", "n_landmarks")},
        "host_stage_seconds_per_step": {k_: round(st[k_], 4) for k_ in ("t_lk", "t_detect", "t_pnp", "t_tri", "t_ba", "t_pnp_kernel", "t_ba_kernel",
                                                                           "t_tri_essential", "t_tri_pose", "tri_hypotheses", "tri_ahead")},
        "trajectory_error_m": {"mean": round(float(terr.mean()), 3), "max": round(float(terr.max()), 3),
                               "travelled": round(float(np.linalg.norm(g[-1])), 1)},
        "input_generation_s": round(t_gen, 2),
    }
    if cpu:
        out["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 2)
        if batched and "value" in batched:
            out["batched_speedup_vs_cpu_baseline"] = round(batched["value"] / cpu["value"], 2)
    if args.config != 1 or args.frames:
        out["note"] = "not the metric configuration: this line is a diagnostic for the named workload"
    kitti_mod = importlib.import_module("practical-multi-view_amd.kitti")
    out["reference_error_report"] = {k_: round(v, 4) for k_, v in kitti_mod.error_report(res.poses, gt0, off).items()}   # OdometryPipeline.cpp:267-296
    if args.poses_out:
        kitti_mod.write_poses_kitti(args.poses_out, res.poses)
    print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
