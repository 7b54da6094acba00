"""Rays and kernel times per depth of a one-chain frame of sphere_scene (RT_FLAG_TIME_DEPTHS): python scripts/gpu_depth_table.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_in_one_weekend_amd as rt
rt.register_default_images()
scene = rt.Scene.build("sphere_scene", 16 / 9)
r = rt.Renderer(0); r.upload(scene)
p = rt.make_params(1920, 1080, 128, max_depth=50, flags=rt._ffi.FLAG_TIME_DEPTHS)
r.render(scene.camera, p)
acc = []
for _ in range(5):
    _, _, st = r.render(scene.camera, p)
    a, b, n = r.depth_timings()
    acc.append((a.copy(), b.copy()))
a = np.median([x[0] for x in acc], axis=0); b = np.median([x[1] for x in acc], axis=0)
print("depth      rays   isect ms  shade ms   ps/ray isect  ps/ray shade   (one chain; device time per ray of the whole chip)")
for d in range(len(a)):
    print(f"{d:5d} {int(n[d]):10d} {a[d]:9.3f} {b[d]:9.3f} {1e9*a[d]/max(n[d],1):12.1f} {1e9*b[d]/max(n[d],1):12.1f}")
print("sum", a.sum(), b.sum(), "depth>=8:", a[8:].sum(), b[8:].sum(), "rays", int(n[8:].sum()), "of", int(n.sum()))
