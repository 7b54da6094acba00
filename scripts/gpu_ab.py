"""Interleaved A/B of library builds in ONE process on ONE device (cdna guide rule 24):
python scripts/gpu_ab.py <spp> <rounds> libA.so libB.so ...   -> median isect/shade/total ms per variant"""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt
from ray_tracing_in_one_weekend_amd import _ffi

spp, rounds = int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:]
rt.register_default_images()
scene = rt.Scene.build(os.environ.get("RTOW_SCENE", "sphere_scene"), 16 / 9)
rends = []
for path in libs:
    _ffi._gpu_lib = None
    _ffi.GPU_LIB_PATH = path
    r = rt.Renderer(0)
    r.upload(scene)
    rends.append(r)
p = rt.make_params(1920, 1080, spp, max_depth=50, flags=rt._ffi.FLAG_TIME_DEPTHS)
res = {i: [] for i in range(len(libs))}
for it in range(rounds + 1):
    for i, r in enumerate(rends):
        img, _, st = r.render(scene.camera, p)
        a, b, n = r.depth_timings()
        if it:  # first round = warm-up
            res[i].append((a.sum(), b.sum(), st.seconds_device * 1e3, st.n_rays))
for i, path in enumerate(libs):
    a = statistics.median(x[0] for x in res[i]); b = statistics.median(x[1] for x in res[i]); t = statistics.median(x[2] for x in res[i])
    print(f"{os.path.basename(path):40s} isect {a:7.2f} ms  shade {b:7.2f} ms  device {t:7.2f} ms  rays {res[i][0][3]}  "
          f"-> {res[i][0][3] / t / 1e3:8.0f} Mray/s")
