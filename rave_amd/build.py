"""Builds librave_hip.so (gfx950) from rave_amd/csrc with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to
the GPU box with the repository snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "librave_hip.so")
SOURCES = ["api.cpp", "pqmf.hip", "pqmf_fold.hip", "pqmf_fold2.hip", "conv_igemm.hip", "conv_igemm_dma.hip", "conv_x6.hip", "conv_x6_i1_121.hip", "conv_x6_i1_221.hip", "conv_x6_i1_321.hip", "conv_x6_i1_122.hip", "conv_x6_i1_222.hip", "conv_x6_i1_322.hip", "conv_x6_i1_211.hip", "conv_x6_i1_311.hip", "conv_x6_i1_212.hip", "conv_x6_i1_312.hip", "conv_x6_i2_121.hip", "conv_x6_i2_221.hip", "conv_x6_i2_321.hip", "conv_x6_i2_122.hip", "conv_x6_i2_222.hip", "conv_x6_i2_322.hip", "conv_x6_i2_211.hip", "conv_x6_i2_311.hip", "conv_x6_i2_212.hip", "conv_x6_i2_312.hip", "conv_x6_i4_121.hip", "conv_x6_i4_221.hip", "conv_x6_i4_321.hip", "conv_x6_i4_122.hip", "conv_x6_i4_222.hip", "conv_x6_i4_322.hip", "conv_x6_i4_211.hip", "conv_x6_i4_311.hip", "conv_x6_i4_212.hip", "conv_x6_i4_312.hip", "unit_x6.hip", "conv_host.hip", "conv_wgrad.hip", "conv_wgrad_x6.hip", "conv_smallc.hip", "conv2d.hip", "conv2d_x6.hip", "conv2d_smallm.hip", "conv2d_smallc.hip", "wgrad2d_x6.hip", "vq.hip", "feed.hip", "misc.hip", "stft_loss.hip", "adam.hip", "feature_match.hip"]
HEADERS = ["common.hpp", "conv_params.hpp", "conv2d_x6.hpp", "conv_x6_kernel.inc", os.path.join("..", "..", "include", "rave_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, _variants: bool = True) -> str:
    """Compile what changed and link the library.  The comparison library under _var/, if it exists, is refreshed in the
    same call (it exports the same ABI: a stale one fails to load in bench.py's forward_only_bf16x6 leg)."""
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hip = _hipcc()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        stamp = obj + ".sha"
        dig = _digest([src] + hdrs)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        lang = ["-x", "hip"] if s.endswith(".hip") or s.endswith(".cpp") else []
        jobs.append((s, [hip] + FLAGS + lang + ["-c", src, "-o", obj], stamp, dig))

    def run(job):
        name, cmd, stamp, dig = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB) or force:
        cmd = [hip, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if _variants and os.path.exists(variant_lib()):
        build_variant(_main_done=True)
    return LIB


# Comparison library (never loaded by the product path: rave_amd._lib takes it only through RAVE_HIP_LIB): the x6 kernels in
# their round 2-5 form -- three bf16 pieces per operand, six partial products, no scales / range slots (common.hpp:
# RH_X6_F16 = 0) -- for bench.py's `forward_only_bf16x6` leg and A/B runs.  Only the sources that see the macro are recompiled.
VARIANT_SOURCES = [s for s in SOURCES if s.startswith("conv_x6_i")] + [
    "conv_x6.hip", "unit_x6.hip", "conv_host.hip", "conv_wgrad_x6.hip", "conv_wgrad.hip", "conv2d_x6.hip", "api.cpp"]
VAR = os.path.join(HERE, "_var")


def variant_lib() -> str:
    return os.path.join(VAR, "librave_hip_bf16.so")


def build_variant(force: bool = False, _main_done: bool = False) -> str:
    if not _main_done:
        build(_variants=False)
    os.makedirs(VAR, exist_ok=True)
    hip = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in VARIANT_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(VAR, f"{s}.bf16.o")
        stamp = obj + ".sha"
        dig = _digest([src] + hdrs) + "-bf16"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((f"{s} [bf16 x 6]", [hip] + FLAGS + ["-DRH_X6_F16=0", "-x", "hip", "-c", src, "-o", obj], stamp, dig))

    def run(job):
        name, cmd, stamp, dig = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    lib = variant_lib()
    objs = [os.path.join(VAR, f"{s}.bf16.o") if s in VARIANT_SOURCES else os.path.join(OBJ, s + ".o") for s in SOURCES]
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        r = subprocess.run([hip, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--variant" in sys.argv or "--variants" in sys.argv:
        print(build_variant(force="--force" in sys.argv))
