"""Build recipes (in-tree, no JIT cache): hipcc for the gfx950 product library, g++ for host-only pieces.

  libpmv_hip.so    HIP kernels + C ABI (include/pmv_hip.h) + host pipeline       -- the product
  libpmv_synth.so  synthetic KITTI-like sequence generator (host only, input data) -- bench/tests input
  oracle/liborc.so CPU restatement of the reference (TEST INFRASTRUCTURE ONLY)   -- built by oracle/Makefile
"""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
    # bit-parity with the oracle: no FMA contraction, IEEE division/sqrt (hipcc default), no fast-math
    "-Wall", "-Wuninitialized", "-Winit-self", "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-value", "-Wno-unused-result",
]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fno-fast-math", "-Wall"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hip_sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip"))) + sorted(glob.glob(os.path.join(HERE, "host", "vo_*.cpp")))


def build_hip(force=False, verbose=False):
    out = os.path.join(HERE, "libpmv_hip.so")
    srcs = hip_sources()
    deps = srcs + glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "host", "*.h")) + \
        [os.path.join(ROOT, "include", "pmv_hip.h")]
    if not force and not _newer(out, deps):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIP_FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-I", os.path.join(HERE, "host")]
    # host .cpp files are compiled as HIP host code too (they call the launchers directly)
    for s in srcs:
        cmd += (["-x", "hip", s] if s.endswith(".cpp") else [s])
    cmd += ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_synth(force=False):
    out = os.path.join(HERE, "libpmv_synth.so")
    src = os.path.join(HERE, "host", "synth.cpp")
    if force or _newer(out, [src]):
        subprocess.check_call(["g++"] + CXX_FLAGS + [src, "-o", out])
    return out


def build_oracle(force=False):
    odir = os.path.join(ROOT, "oracle")
    if force:
        subprocess.check_call(["make", "-C", odir, "clean"])
    subprocess.check_call(["make", "-C", odir, "-s"])
    return os.path.join(odir, "liborc.so")


def build_all(force=False, verbose=False):
    return {"hip": build_hip(force, verbose), "synth": build_synth(force), "oracle": build_oracle(force)}


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose=True))
