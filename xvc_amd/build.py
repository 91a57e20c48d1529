"""Builds libxvcgpu.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m xvc_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libxvcgpu.so")
SOURCES = ["xvcgpu.hip", "xvcgpu_comm.hip", "xvcgpu_tables.cpp"]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # the few double-precision metric steps must evaluate op-for-op like the
    # reference's x86-64 build (no FMA contraction); integer code is unaffected
    "-ffp-contract=off",
    "-Wall", "-Wno-unused-function", "-Wno-unused-value",
]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def deps():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    inc = os.path.join(os.path.dirname(HERE), "include")
    files += [os.path.join(inc, f) for f in os.listdir(inc)]
    return files


HOST_OUT = os.path.join(HERE, "libxvchost.so")
HOST_SOURCES = ["xvc_picture_decoder.cc", "xvc_picture_schedule.cc", "xvc_inter_search.cc",
                "xvc_shard_filter.cc", "xvc_cu_state.cc", "xvc_cu_state_builder.cc", "xvc_shard_engine.cc", "xvc_picture_engine.cc"]


def build_host(force=False, verbose=False):
    """libxvchost.so: the C++ host layer above the C-ABI (xvc_amd/host/*.cc:
    picture-level drivers named after the reference's classes), plain g++,
    linked against libxvcgpu.so next to it."""
    host = os.path.join(HERE, "host")
    inc = os.path.join(os.path.dirname(HERE), "include")
    srcs = [os.path.join(host, f) for f in HOST_SOURCES]
    dep = srcs + [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".h")] + \
        [os.path.join(inc, f) for f in os.listdir(inc)]
    if not force and os.path.exists(HOST_OUT):
        m = os.path.getmtime(HOST_OUT)
        if all(os.path.getmtime(f) <= m for f in dep):
            return HOST_OUT
    cmd = ["g++", "-std=c++11", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", inc,
           "-I", host] + srcs + ["-o", HOST_OUT, "-L", HERE, "-lxvcgpu", "-lpthread", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_OUT


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT):
        m = os.path.getmtime(OUT)
        if all(os.path.getmtime(f) <= m for f in deps()):
            return OUT
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
