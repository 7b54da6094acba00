"""(ROUND 3 RECORD: the reorder what-if needs the RT_WHATIF_REORDER build of scripts/experiments/whatif_switches_and_reorder_pass.patch, whose
library still reads RTOW_WHATIF_REORDER; the product library reads no environment variable.  `treemem` uses rt_debug_set_option.)
Round-3 what-if probes on one GPU, one process (interleaved, same-binary switches where possible):
    python scripts/gpu_r3_probe.py reorder  <lib_reorder.so> [spp] [rounds] [scene]   secondary-ray coherence what-if (RTOW_WHATIF_REORDER)
    python scripts/gpu_r3_probe.py lanes    <lib_reorder_lanes.so> [spp] [scene]      lane statistics of the same, + the Perlin section at depth 0
    python scripts/gpu_r3_probe.py treemem  <lib.so> [spp] [rounds] [scene]           tree in LDS against the same tree read through L2 (RTOW_BVH_HBM)
Frames of every variant are compared bit for bit with the first one."""
import ctypes
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt
from ray_tracing_in_one_weekend_amd import _ffi

BOX = {"sphere_scene": "-16,-0.5,-16,32,4,32", "pbr_sweep_scene": "-16,-0.5,-16,32,4,32", "cornell_box": "0,0,0,555,555,555",
       "final_scene": "-200,0,-200,800,600,800"}
# oct, bits per axis, dims, order (0 octant-major, 1 cell-major)
MODES = [("queue order (no reorder)", None), ("octant", "1,0,3,0"), ("octant > 3D cell 8^3", "1,3,3,0"), ("3D cell 8^3 > octant", "1,3,3,1"),
         ("3D cell 16^3", "0,4,3,0"), ("2D cell 16^2 > octant", "1,4,2,1"), ("octant > 2D cell 16^2", "1,4,2,0"), ("2D cell 64^2", "0,6,2,0"),
         ("queue order again", None)]


def load(path):
    _ffi._gpu_lib = None
    _ffi.GPU_LIB_PATH = path
    return rt.Renderer(0)


def lane_stats(reset=True):
    lib = _ffi.load_gpu_library()
    fn = lib.rt_debug_lane_stats
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 24)()
    assert fn(out, 1 if reset else 0) == 0
    return list(out)


NAMES = ["main loop (lanes holding a ray)", "node step", "leaf loop trip", "refill block (lanes refilled)", "ball sampler: loop trip",
         "ball sampler: call", "depth-0 list test: trip", "depth-0: waves / lanes with a list", "Perlin turbulence: waves entering / lanes",
         "k_shade: 64-ray segments / lanes with a ray"]


def print_stats(out, n_rays, only=None):
    for i, n in enumerate(NAMES):
        if only and i not in only:
            continue
        slots, lanes = out[2 * i], out[2 * i + 1]
        print(f"    {n:44s} wave-trips {slots // 64:12d}  active lanes {lanes:14d}  utilisation {lanes / max(slots, 1):.3f}  per ray {lanes / n_rays:.3f}")


what = sys.argv[1]
path = sys.argv[2]
rt.register_default_images()
if what == "reorder":
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    name = sys.argv[5] if len(sys.argv) > 5 else "sphere_scene"
    maxd = os.environ.get("RTOW_REORDER_MAXDEPTH", "50")
    square = name in ("cornell_box", "final_scene")
    scene = rt.Scene.build(name, 1.0 if square else 16 / 9)
    r = load(path)
    r.upload(scene)
    p = rt.make_params(1080 if square else 1920, 1080, spp, max_depth=50, flags=_ffi.FLAG_TIME_DEPTHS)
    res = {i: [] for i in range(len(MODES))}
    ref = None
    for it in range(rounds + 1):
        for i, (label, mode) in enumerate(MODES):
            if mode:
                os.environ["RTOW_WHATIF_REORDER"] = f"{mode},{maxd},{BOX[name]}"
            else:
                os.environ.pop("RTOW_WHATIF_REORDER", None)
            img, _, st = r.render(scene.camera, p)
            a, b, n = r.depth_timings()
            if ref is None:
                ref = img.copy()
            assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), f"frame differs: {label}"
            if it:
                res[i].append((a.copy(), b.copy(), st.seconds_device * 1e3))
    print(f"# {name} 1920x1080 x {spp} spp, single chain (per-depth timing), median of {rounds}; k_whatif_reorder runs UNTIMED in front of "
          f"k_intersect at depths 1..{maxd}; frames bit-identical to queue order (asserted)")
    print(f"{'order of the rays in each shard':34s} {'isect all':>9s} {'d1':>7s} {'d2':>7s} {'d3':>7s} {'d4+':>7s} | {'shade d>=1':>10s} {'d1':>7s}")
    for i, (label, mode) in enumerate(MODES):
        a = np.median([x[0] for x in res[i]], axis=0)
        b = np.median([x[1] for x in res[i]], axis=0)
        print(f"{label:34s} {a.sum():9.3f} {a[1]:7.3f} {a[2]:7.3f} {a[3]:7.3f} {a[4:].sum():7.3f} | {b[1:].sum():10.3f} {b[1]:7.3f}")
elif what == "lanes":
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    name = sys.argv[4] if len(sys.argv) > 4 else "sphere_scene"
    square = name in ("cornell_box", "final_scene")
    scene = rt.Scene.build(name, 1.0 if square else 16 / 9)
    r = load(path)
    r.upload(scene)
    nx = 1080 if square else 1920
    for label, mode in MODES[:-1]:
        if mode:
            os.environ["RTOW_WHATIF_REORDER"] = f"{mode},50,{BOX[name]}"
        else:
            os.environ.pop("RTOW_WHATIF_REORDER", None)
        lane_stats()
        _, _, st = r.render(scene.camera, rt.make_params(nx, 1080, spp, max_depth=50))
        print(f"## {name}, {spp} spp, {st.n_rays} rays, shard order: {label}")
        print_stats(lane_stats(), st.n_rays, only=(0, 1, 2, 3))
    os.environ.pop("RTOW_WHATIF_REORDER", None)
    lane_stats()
    _, _, st = r.render(scene.camera, rt.make_params(nx, 1080, spp, max_depth=0))
    print(f"## {name}, {spp} spp, depth 0 only (max_depth 0): {st.n_rays} primary rays")
    print_stats(lane_stats(), st.n_rays, only=(4, 5, 6, 7, 8, 9))
    lane_stats()
    _, _, st = r.render(scene.camera, rt.make_params(nx, 1080, spp, max_depth=50))
    print(f"## {name}, {spp} spp, all depths: {st.n_rays} rays")
    print_stats(lane_stats(), st.n_rays, only=(4, 5, 8, 9))
elif what == "treemem":
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    name = sys.argv[5] if len(sys.argv) > 5 else "sphere_scene"
    scene = rt.Scene.build(name, 16 / 9)
    rends = []
    for label, env in (("tree + geometry in LDS", None), ("tree + geometry read through L2 (tree_placement = 1)", "1"), ("LDS again", None)):
        r = load(path) if not rends else rt.Renderer(0)
        r.set_option("grid", 1)  # the tree at every depth
        if env:
            r.set_option("tree_placement", 1)
        r.upload(scene)
        rends.append((label, r))
    p = rt.make_params(1920, 1080, spp, max_depth=50, flags=_ffi.FLAG_TIME_DEPTHS)
    res = {i: [] for i in range(len(rends))}
    for it in range(rounds + 1):
        for i, (label, r) in enumerate(rends):
            _, _, st = r.render(scene.camera, p)
            a, b, n = r.depth_timings()
            if it:
                res[i].append((a.sum(), b.sum(), st.seconds_device * 1e3))
    print(f"# {name} 1920x1080 x {spp} spp, per-depth timing (single chain), median of {rounds}")
    for i, (label, r) in enumerate(rends):
        print(f"{label:52s} isect {statistics.median(x[0] for x in res[i]):8.3f} ms  shade {statistics.median(x[1] for x in res[i]):8.3f} ms  "
              f"device {statistics.median(x[2] for x in res[i]):8.3f} ms")
