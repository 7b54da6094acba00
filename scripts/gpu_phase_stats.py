"""Where a wave of the class-sorting k_shade spends its time, from a -DRT_PROFILE_PHASES build (rt_kernels.h: RT_PHASE_CLOCK):
python scripts/gpu_phase_stats.py build/librtow_phases.so [spp] [scene ...]
(the diagnostic build drains the first segment's gather on its own, so its frame is slower than the product's; the split is the point)"""
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
for name in scenes:
    scene = rt.Scene.build(name, 16 / 9)
    r = rt.Renderer(0)
    r.upload(scene)
    lib = _ffi.load_gpu_library()
    fn = lib.rt_debug_phase_stats
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 16)()
    p = rt.make_params(1920, 1080, spp, max_depth=50)
    r.render(scene.camera, p)
    assert fn(out, 1) == 0
    _, _, st = r.render(scene.camera, p)
    assert fn(out, 1) == 0
    total = max(out[0], 1)
    print(f"{name}: {st.n_rays} rays, {spp} spp, k_shade past depth 0: {total} wave-ticks (s_memtime) in all")
    rows = [("drain of the previous block's stores (vmcnt(0) on its own)", out[8], out[5], "blocks"),
            ("  of the sort: hit-record loads issued -> arrived", out[9], out[5], "blocks"),
            ("sort of a block (hit-record loads, histogram, scatter)", out[1], out[5], "blocks"),
            ("wait for the first segment's rays", out[2], out[5], "blocks"),
            ("all-miss segments", out[3], out[6], "segments"),
            ("segments with hits", out[4], out[7], "segments")]
    for label, ticks, n, unit in rows:
        print(f"  {label:56s} {100.0 * ticks / total:5.1f} %   {n:10d} {unit:8s} {ticks / max(n, 1):9.0f} ticks each")
    print(f"  {'outside (staging, barriers, tails)':56s} {100.0 * (total - out[1] - out[2] - out[3] - out[4] - out[8]) / total:5.1f} %")
