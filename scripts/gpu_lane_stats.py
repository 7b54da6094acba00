"""Lane utilisation inside k_intersect's loops, from a -DRT_PROFILE_LANES build (rt_kernels.h: RT_LANE_STAT):
python scripts/gpu_lane_stats.py build/lib_lanes.so [spp] [scene ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt
from ray_tracing_in_one_weekend_amd import _ffi

path = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scenes = sys.argv[3:] or ["sphere_scene"]
_ffi._gpu_lib = None
_ffi.GPU_LIB_PATH = path
rt.register_default_images()
names = ["main loop (lanes holding a ray)", "node step", "leaf loop trip", "refill block (lanes refilled)",
         "random_in_unit_sphere: loop trip", "random_in_unit_sphere: call", "depth-0 list test: trip", "depth-0: wave / lanes with a list",
         "Perlin turbulence: waves entering / lanes", "k_shade: 64-ray segments / lanes with a ray",
         "shade(): Emission", "shade(): Diffuse", "shade(): Lambert + pbr.rs", "shade(): Metal", "shade(): Dielectric"]
for name in scenes:
    scene = rt.Scene.build(name, 16 / 9)
    r = rt.Renderer(0)
    r.upload(scene)
    lib = _ffi.load_gpu_library()
    fn = lib.rt_debug_lane_stats
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 48)()
    p = rt.make_params(1920, 1080, spp, max_depth=50)
    assert fn(out, 1) == 0
    _, _, st = r.render(scene.camera, p)
    assert fn(out, 1) == 0
    print(f"{name}: {st.n_rays} rays, {spp} spp (depth >= 1 for the main loop and the refill; all depths for the steps)")
    for i, n in enumerate(names):
        slots, lanes = out[2 * i], out[2 * i + 1]
        print(f"  {n:36s} wave-trips {slots // 64:12d}  active lanes {lanes:14d}  utilisation {lanes / max(slots, 1):.3f}"
              f"  per ray {lanes / st.n_rays:.2f}")
