"""Builds the native pieces in-tree with hipcc (gfx950 only) and g++.

  ganon_amd/csrc/libganon_hip.so   HIP kernels + C ABI (include/ganon_hip.h)
  ganon_amd/host/ganon-classify    C++ host binary (drop-in CLI), links libganon_hip.so
  ganon_amd/host/ganon-build       filter writer, ganon_amd/host/ganon-reassign  the EM over .all

No JIT cache: the .so / binary live next to their sources so they travel with a repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(CSRC, "libganon_hip.so")
BIN = os.path.join(HOST, "ganon-classify")
BIN_BUILD = os.path.join(HOST, "ganon-build")
BIN_REASSIGN = os.path.join(HOST, "ganon-reassign")
BUILD_ONLY = ("build.cpp", "build_params.cpp")  # sources of ganon-build that ganon-classify does not link
REASSIGN_ONLY = ("reassign.cpp", "reassign_main.cpp")  # ganon-reassign (the EM over .all)

HIP_SOURCES = ["gn_kernels.hip", "gn_split.hip", "gn_minimiser_lpr.hip", "gn_hibf.hip", "gn_postfilter.hip", "gn_build.hip", "gn_gather.hip", "gn_fastq.hip", "gn_reassign.hip", "gn_inflate.hip", "gn_capi.hip"]
HIP_HEADERS = ["gn_internal.h", "gn_scan.h", os.path.join(ROOT, "include", "ganon_hip.h"), os.path.join(ROOT, "include", "ganon_ibf_hash.h")]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libganon_hip.so cannot be built (there is no CPU fallback)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """one object per .hip source (compiled in parallel, only when stale), then one link"""
    import concurrent.futures as cf
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = None
    jobs, objs = [], []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            hipcc = hipcc or _hipcc()
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
                         "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc or _hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs)
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    """ganon-classify (every host source but the builder's) and ganon-build (its own sources + the sequence reader)"""
    if not os.path.isdir(HOST):
        return ""
    every = sorted(f for f in os.listdir(HOST) if f.endswith(".cpp"))
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "ganon_hip.h"), os.path.join(ROOT, "include", "ganon_ibf_hash.h")]
    build_hip(force=False, verbose=verbose)
    for binary, names in ((BIN, [f for f in every if f not in BUILD_ONLY + REASSIGN_ONLY]),
                          (BIN_BUILD, [f for f in every if f in BUILD_ONLY] + ["seq_io.cpp", "pgzip.cpp"]),
                          (BIN_REASSIGN, [f for f in every if f in REASSIGN_ONLY])):
        srcs = [os.path.join(HOST, f) for f in names]
        if not srcs:
            continue
        if force or _stale(binary, srcs + hdrs + [LIB]):
            cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", binary] + srcs + \
                  ["-L", CSRC, "-lganon_hip", "-lz", "-ldl", "-Wl,-rpath,$ORIGIN/../csrc"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    return BIN


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
