"""Batched multi-sequence path ("same kernels with a leading batch dimension"): B independent sequences through
pmv_pipeline_run_batch — one k_lk_batch / detector / k_pnp_*_batch / k_bamB_* / k_tri_dlt_batch launch per kernel class for the
requests of all sequences — must give, for every sequence, exactly the bits of that sequence's own pmv_pipeline_run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K00 = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)


def _stage(pmv, ctx, cfg, lengths, seeds):
    seqs, first, data = [], 0, []
    for n, seed in zip(lengths, seeds):
        frames, gt = pmv.synth_sequence(seed, 0, n, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], nthreads=16)
        ctx.frames_stage(first, frames)
        seqs.append((first, n, gt))
        data.append((frames, gt))
        first += n
    return seqs, data


def _assert_same(a, b, what):
    assert np.array_equal(a.poses, b.poses), f"{what}: poses differ"
    assert len(a.features) == len(b.features)
    for k, (x, y) in enumerate(zip(a.features, b.features)):
        assert np.array_equal(x, y), f"{what}: features of frame {k} differ"
    for key in ("lk_calls", "lk_points", "detect_calls", "pnp_calls", "pnp_points", "tri_calls", "ba_calls", "ba_obs", "ba_points", "init_offset"):
        assert a.stats[key] == b.stats[key], (what, key)


def test_batch_of_different_sequences_equals_single_runs(pmv, gpu_ctx_factory):
    """six sequences of different lengths and content (metric configuration: 400 tracks, bundle 5) in one batch"""
    cfg = K00
    lengths = [60, 41, 75, 33, 52, 60]
    seeds = [1000, 1001, 1002, 1003, 1004, 1000]     # the last one repeats the first: same input, different batch slot
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=sum(lengths), max_tracks=4096)
    seqs, data = _stage(pmv, ctx, cfg, lengths, seeds)
    ctx.lk_counters(reset=True)
    got = ctx.pipeline_run_batch(seqs, cfg["w"], cfg["h"], K)
    lk_work_batch = ctx.lk_counters()
    st = ctx.batch_stats()
    print("combiner statistics:", st)
    for role in ("lk", "pnp", "ba"):
        assert st[role]["requests"] > st[role]["launches"] > 0, f"{role}: nothing was merged into a shared launch"
    single = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=max(lengths), max_tracks=4096)
    single.lk_counters(reset=True)
    for b, (frames, gt) in enumerate(data):
        single.frames_stage(0, frames)
        ref = single.pipeline_run(lengths[b], cfg["w"], cfg["h"], K, gt, threaded=1)
        _assert_same(got[b], ref, f"sequence {b}")
    # the measured LK work (iterations, level passes, tracks: the roofline's OPS_lk) of the batched launches = that of the single runs
    assert lk_work_batch == single.lk_counters() and lk_work_batch[2] > 0
    _assert_same(got[0], got[5], "same input in two batch slots")
    # a second batch on the same engine, in another order and with a different batch size
    again = ctx.pipeline_run_batch([seqs[3], seqs[1], seqs[4]], cfg["w"], cfg["h"], K)
    for r, b in zip(again, (3, 1, 4)):
        _assert_same(r, got[b], f"re-run of sequence {b}")


def test_batch_with_one_workgroup_ba_equals_single_runs_in_that_mode(pmv, gpu_ctx_factory):
    """pmv_set_ba_mode(1): the BA combiner solves every problem of a round in one launch (k_ba_lm_batch, one workgroup per problem);
    a sequence's results equal its own run with the same mode, bit for bit, and stay within the BA bar of the default mode."""
    cfg = K00
    lengths = [48, 37, 55, 48]
    seeds = [1000, 1001, 1002, 1000]
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=sum(lengths), max_tracks=4096)
    ctx.set_ba_mode(1)
    seqs, data = _stage(pmv, ctx, cfg, lengths, seeds)
    got = ctx.pipeline_run_batch(seqs, cfg["w"], cfg["h"], K)
    st = ctx.batch_stats()
    assert st["ba"]["requests"] > st["ba"]["launches"] > 0
    single = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=max(lengths), max_tracks=4096)
    single.set_ba_mode(1)
    for b, (frames, gt) in enumerate(data):
        single.frames_stage(0, frames)
        ref = single.pipeline_run(lengths[b], cfg["w"], cfg["h"], K, gt, threaded=1)
        _assert_same(got[b], ref, f"sequence {b} (BA mode 1)")
    _assert_same(got[0], got[3], "same input in two batch slots")
    single.set_ba_mode(0)
    single.frames_stage(0, data[1][0])
    ref0 = single.pipeline_run(lengths[1], cfg["w"], cfg["h"], K, data[1][1], threaded=1)
    # the two BA forms differ in summation order only: the trajectories agree far below the drift of the method
    d = np.abs(np.asarray(got[1].poses) - np.asarray(ref0.poses)).max()
    print("largest pose difference between the two BA forms over", lengths[1], "frames:", d)
    assert d < 1e-2


def test_batch_other_configurations(pmv, gpu_ctx_factory):
    """ShiTomasi extractor, and 800 tracks / bundle 10 (60x60 reduced camera system): batched == single"""
    cfg = K00
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=130, max_tracks=4096)
    seqs, data = _stage(pmv, ctx, cfg, [30, 26, 38, 34], [1005, 1006, 1007, 1008])
    single = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=40, max_tracks=4096)
    for kw in (dict(extractor=1), dict(min_tracked=800, tol=300, bundle_size=10)):
        got = ctx.pipeline_run_batch(seqs, cfg["w"], cfg["h"], K, **kw)
        for b, (frames, gt) in enumerate(data):
            single.frames_stage(0, frames)
            ref = single.pipeline_run(seqs[b][1], cfg["w"], cfg["h"], K, gt, threaded=1, **kw)
            _assert_same(got[b], ref, f"{kw}: sequence {b}")


def test_batch_argument_errors(pmv, gpu_ctx_factory):
    cfg = K00
    K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
    ctx = gpu_ctx_factory(cfg["w"], cfg["h"], n_slots=20, max_tracks=1024)
    gt = np.zeros((30, 12))
    with pytest.raises(pmv.PmvError) as e:
        ctx.pipeline_run_batch([(0, 30, gt)], cfg["w"], cfg["h"], K)          # more frames than slots
    assert e.value.code == -3
    with pytest.raises(pmv.PmvError) as e:
        ctx.pipeline_run_batch([(0, 10, gt[:10])], cfg["w"], cfg["h"], K)      # slots never staged
    assert e.value.code in (-2, -4)
