"""The C-ABI library loads and exports every function include/pmv_hip.h declares (no compute without a GPU), and creating
a context without a gfx950 device fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "pmv_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pmv_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_all_exported(pmv):
    lib = pmv.load_library()
    names = _declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/pmv_hip.h but not exported: {missing}"
    # and the Python binding's list covers the header
    assert sorted(set(pmv.ABI_SYMBOLS)) == names, sorted(set(pmv.ABI_SYMBOLS) ^ set(names))


def test_product_library_does_not_link_the_oracle(pmv):
    out = subprocess.check_output(["ldd", pmv.lib_path()], text=True)
    assert "liborc" not in out
    syms = subprocess.check_output(["nm", "-D", "--defined-only", pmv.lib_path()], text=True)
    assert " orc_" not in syms


def test_context_creation_fails_without_gpu(pmv):
    import glob
    if glob.glob("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(pmv.PmvError) as e:
        pmv.Context(640, 480)
    assert e.value.code in (-1, -4)
    assert "fallback" in str(e.value) or "device" in str(e.value).lower()


def test_oracle_library_loads(orc):
    for sym in ("orc_pyr_down", "orc_scharr", "orc_lk_track", "orc_gftt_cell", "orc_shitomasi_cell", "orc_ba_residuals", "orc_ba_solve",
                "orc_pnp_ransac", "orc_pipeline_run"):
        assert hasattr(orc.lib, sym)


def test_stats_buffer_size_matches_header_and_binding(pmv):
    """pmv_pipeline_get_stats writes pmv_pipeline_stats_count() doubles: the header documents that number and every field,
    the Python binding names every field (a C caller that sizes its buffer from the header must not be overrun)."""
    lib = pmv.load_library()
    n = lib.pmv_pipeline_stats_count()
    src = open(os.path.join(ROOT, "include", "pmv_hip.h")).read()
    doc = src[src.index("Run statistics"):src.index("int pmv_pipeline_stats_count")]
    assert f"(= {n})" in doc
    fields = [int(k) for k in re.findall(r"\[(\d+)\]", doc)]
    assert fields == list(range(n)), "the header must document every statistics slot exactly once, in order"
    assert len(pmv.STAT_KEYS) == n
    assert re.search(r"pmv_pipeline_get_stats\([^)]*double\* out%d\)" % n, src)
    # the oracle's twin reports the same count (same host code)
    import orc_binding
    assert orc_binding.load().lib.orc_pipeline_stats_count() == n


def test_pipeline_params_layout_matches_header(pmv):
    """ctypes mirror of pmv_pipeline_params: same field names in the same order as the header"""
    src = open(os.path.join(ROOT, "include", "pmv_hip.h")).read()
    body = src[src.index("typedef struct pmv_pipeline_params {"):src.index("} pmv_pipeline_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"int ([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [f[0] for f in pmv.PipelineParams._fields_]
