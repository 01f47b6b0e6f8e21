"""Builds libnesr_hip.so in-tree with hipcc for gfx950 (MI355X).

``python -m neural_enhanced_super_resolution_amd.build`` or ``build_library()``.
The .so is git-ignored but travels with the tree (it must exist before the GPU box runs
anything: there is no JIT fallback and no CPU fallback).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libnesr_hip.so")
SOURCES = ["conv3x3_mfma.hip", "conv3x3_bf16.hip", "conv3x3_wino_f32.hip", "conv3x3_f16x2.hip", "rdb_bf16_strip.hip", "imgproc.hip", "pack.hip", "nesr_api.cpp"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(objs_src):
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = list(objs_src) + [os.path.join(CSRC, "nesr_kernels.h"),
                             os.path.join(HERE, "..", "include", "nesr_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, extra_flags=()):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and not _stale(srcs):
        return LIB_PATH
    cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-I", CSRC, *extra_flags, "-o", LIB_PATH + ".tmp"]
    for s in srcs:
        cmd += ["-x", "hip", s]
    if verbose:
        print("[nesr build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB_PATH)
