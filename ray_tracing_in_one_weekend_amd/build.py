"""Build recipes for the native libraries (in-tree, so the built .so travels with the repo
snapshot to the GPU box).  hipcc cross-compiles gfx950 without a GPU."""
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)

# -fno-slp-vectorize: the SLP vectoriser pairs the x/y/z arithmetic of the V3 helpers into v_pk_mul_f32 / v_pk_add_f32,
# which issue at half the rate of the scalar forms on gfx950 (same lane-ops per cycle, profiles/round2/valu_peak.json)
# but need aligned register pairs and v_pk_mov shuffles: k_shade 6 % and k_intersect 1 % faster without (same bits).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


def build_gpu_debug_library(out=None):
    """The same library with -DRT_DEBUG_QUEUE_BOUNDS (device-side traps on a queue position beyond its shard's capacity);
    select it with RTOW_GPU_LIB=<path> for a test run."""
    csrc = os.path.join(PKG_DIR, "csrc")
    out = out or os.path.join(ROOT, "build", "librtow_mi355x_debug.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    _run([hipcc] + HIPCC_FLAGS + ["-DRT_DEBUG_QUEUE_BOUNDS", "-o", out, os.path.join(csrc, "rt_api.hip")])
    return out


def gpu_sources():
    csrc = os.path.join(PKG_DIR, "csrc")
    return [os.path.join(csrc, f) for f in ("rt_api.hip", "rt_kernels.h", "rt_device.h", "rt_bvh.h", "rt_grid.h", "rt_multi.h", "rt_pool.h")] + \
           [os.path.join(ROOT, "include", "rtow_mi355x.h")]


def gpu_build_id():
    """16 hex digits over the device sources and the compiler flags: what rt_build_id() of a library built from this tree returns.
    Profiles under profiles/ record it, and bench.py refuses to quote a profile taken on other kernels."""
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for s in gpu_sources():
        h.update(open(s, "rb").read())
    return h.hexdigest()[:16]


def build_gpu_library(force=False):
    """hipcc --offload-arch=gfx950: HIP kernels + C-ABI -> librtow_mi355x.so"""
    srcs = gpu_sources()
    out = os.path.join(PKG_DIR, "librtow_mi355x.so")
    if force or _newer(out, srcs):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        _run([hipcc] + HIPCC_FLAGS + [f'-DRT_BUILD_ID="{gpu_build_id()}"', "-o", out, srcs[0]])
    return out


def build_host_library(force=False):
    """g++: C++ mirror of the reference construction API + C handle API -> librtow_host.so"""
    host = os.path.join(PKG_DIR, "host")
    srcs = [os.path.join(host, f) for f in ("demo_scene.cpp", "host_capi.cpp", "png_out.cpp", "rtow.hpp")]
    srcs += [os.path.join(ROOT, "include", f) for f in ("rtow_mi355x.h", "rtow_host.h")]
    out = os.path.join(PKG_DIR, "librtow_host.so")
    if force or _newer(out, srcs):
        _run(["g++"] + HOST_FLAGS + ["-o", out, srcs[0], srcs[1], srcs[2], "-lz"])
    return out


def build_c_host_example(force=False):
    """gcc -std=c99: examples/host_main.c, a host through nothing but include/*.h (the C stand-in for the reference's main.rs:62-129),
    linked against the two libraries beside it -> build/host_main"""
    src = os.path.join(ROOT, "examples", "host_main.c")
    out = os.path.join(ROOT, "build", "host_main")
    deps = [src] + [os.path.join(ROOT, "include", f) for f in ("rtow_mi355x.h", "rtow_host.h")]
    if force or _newer(out, deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"), "-o", out, src, "-L", PKG_DIR,
              "-l:librtow_host.so", "-l:librtow_mi355x.so", "-Wl,-rpath,$ORIGIN/../ray_tracing_in_one_weekend_amd"])
    return out


def build_all(force=False):
    """The product libraries only.  The CPU oracle is test infrastructure and builds from its own directory (oracle/build.py)."""
    return [build_gpu_library(force), build_host_library(force), build_c_host_example(force)]
