"""throughput of pmv_pipeline_run_batch for several batch sizes (diagnostic; bench.py reports the chosen one)
usage: python scripts/bench_batch.py [B ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
if os.environ.get("PIN"):
    cpus = sorted(os.sched_getaffinity(0))[:int(os.environ["PIN"])]
    os.sched_setaffinity(0, cpus)
    print("pinned to CPUs", cpus, flush=True)
pmv = importlib.import_module("practical-multi-view_amd")
cfg = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
n = int(os.environ.get("FRAMES", "1101"))
Bs = [int(x) for x in sys.argv[1:]] or [8, 16, 32]
frames, gt = pmv.synth_sequence(1007, 0, n, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=16)
DISTINCT = int(os.environ.get("DISTINCT", "1"))   # >1: bench.py's layout - 4 seeds x start offsets 0/40/80/120, cycled over the slots
distinct = [(frames, gt)]
if DISTINCT > 1:
    OFF, distinct = 40, []
    for k in range((DISTINCT + 3) // 4):
        f_, g_ = pmv.synth_sequence(1007 + 64 + k, 0, n + 3 * OFF, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=16)
        distinct += [(f_[OFF * d: OFF * d + n], g_[OFF * d: OFF * d + n]) for d in range(4)]
    distinct = distinct[:DISTINCT]
K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
Bmax = max(Bs)
ctx = pmv.Context(cfg["w"], cfg["h"], n_slots=Bmax * n, max_tracks=1024, max_ba_cams=8, max_ba_points=2048, max_ba_obs=16384)
for b in range(Bmax):
    ctx.frames_stage(b * n, distinct[b % len(distinct)][0])
if os.environ.get("BA_MODE"):
    ctx.set_ba_mode(int(os.environ["BA_MODE"]))   # 1: one workgroup per solve, one launch per batched BA round
ref = None
THREADED = int(os.environ.get("THREADED", "1"))
DEVFP = int(os.environ.get("DEVFP", "0"))
BA_IT = int(os.environ.get("BA_ITERS", "5"))   # diagnostic: how sensitive is the throughput to the length of the BA launch chain
print("wait mode", os.environ.get("PMV_BATCH_WAIT", "flag"), "threaded", THREADED, flush=True)
for B in Bs:
    seqs = [(b * n, n, distinct[b % len(distinct)][1]) for b in range(B)]
    r = ctx.pipeline_run_batch(seqs, cfg["w"], cfg["h"], K, want_features=False, defer_free=True, threaded=THREADED, device_fivepoint=DEVFP, ba_iterations=BA_IT)   # warm-up
    ctx.sync()
    s0 = ctx.batch_stats()
    t0 = time.perf_counter()
    c0 = time.process_time()
    r2 = ctx.pipeline_run_batch(seqs, cfg["w"], cfg["h"], K, want_features=False, defer_free=True, threaded=THREADED, device_fivepoint=DEVFP, ba_iterations=BA_IT)
    ctx.sync()
    dt = time.perf_counter() - t0
    cpu = time.process_time() - c0
    s1 = ctx.batch_stats()
    fr = sum(n - int(x.stats["init_offset"]) for x in r2)
    if ref is None:
        ref = r2[0].poses
    same = all(np.array_equal(x.poses, r2[i % len(distinct)].poses) for i, x in enumerate(r2))   # (every copy of a distinct sequence gives the same poses)
    print(f"B={B:3d}: {fr / dt:9.1f} frames/s  ({dt:.3f} s)  identical={same}  process CPU {cpu:.2f} s = {cpu / fr * 1e6:.0f} us per frame, {cpu / dt:.1f} cores busy", flush=True)
    keys = [k for k in ("t_lk", "t_detect", "t_pnp", "t_tri", "t_ba", "t_tri_essential", "t_tri_pose", "tri_hypotheses", "tri_ahead") if k in r2[0].stats]
    print("        per frame (wall of the calling threads, us):", {k: round(sum(float(x.stats[k]) for x in r2) / fr * (1.0 if k.startswith("tri_") else 1e6), 2) for k in keys}, flush=True)
    for role in s1:
        d = {k: s1[role][k] - s0[role][k] for k in s1[role]}
        if d["launches"]:
            print(f"        {role:4s} {d['requests'] / d['launches']:5.1f} req/launch, {d['launches']:5d} launches, cpu {d['cpu_s']:.2f} s, work {d['work_s']:.2f} s (sync {d['sync_s']:.2f} s)"
                  f"  -> {d['work_s'] / d['launches'] * 1e6:.0f} us per round", flush=True)
    if os.environ.get("PMV_LK_STAMPS"):   # phase timers of every 64th track of the batched LK launches (shader cycles)
        import ctypes as C
        out = np.zeros(16, np.uint64)
        ctx.lib.pmv_debug_lk_stamps(ctx.h, out.ctypes.data_as(C.POINTER(C.c_uint64)))
        nt, it = max(1, int(out[13])), max(1, int(out[8]))
        names = ["level-entry", "I-tile", "scharr", "samples+A", "iterations(+J tiles)", "err-pass"]
        print("        LK stamps:", nt, "sampled tracks; cycles per track:", " ".join("%s=%d" % (nm, int(out[i]) / nt) for i, nm in enumerate(names)), "iterations/track %.1f" % (it / nt), "| tiles waited for in place per track %.2f, wall %.1f us per track (100 MHz counter)" % (int(out[14]) / nt, int(out[15]) / nt / 100.0))
        print("        per iteration: top(tile check / J stage)=%d sample+diff=%d wave-sum=%d update=%d cycles" % tuple(int(out[k]) / it for k in (9, 10, 11, 12)), flush=True)
    for x in r + r2:
        x.free()
ctx.close()
