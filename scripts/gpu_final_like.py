"""A final_scene-like general scene of adjustable size (demo_scene.rs:150-221: box field of GBoxes, an instanced cloud of spheres
under Translate(RotateY(..)), a light, a few big spheres) rendered with its tree in LDS and with the same tree read through
L2 (rt_debug_set_option tree_placement = 1, as the real final_scene must): what would final_scene gain from a tree that fits LDS?
    python scripts/gpu_final_like.py <boxes_per_side> <cloud_spheres> [spp] [rounds]"""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt
from ray_tracing_in_one_weekend_amd import _ffi

nb, nc = int(sys.argv[1]), int(sys.argv[2])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.default_rng(1995)
s = rt.Scene.new()
f = _ffi
ground = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.48, 0.83, 0.53)))
white = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.73, 0.73, 0.73)))
light = s.material(f.MAT_EMISSION, tex0=s.constant_tex((7, 7, 7)))
w = 2000.0 / nb
for i in range(nb):
    for j in range(nb):
        x0, z0 = -1000.0 + i * w, -1000.0 + j * w
        s.gbox((x0, 0.0, z0), (x0 + w, float(rng.random() * 100 + 1), z0 + w), ground)
s.rect(f.RECT_XZ if hasattr(f, "RECT_XZ") else 1, (123, 544, 147), (423, 554, 412), light)
for k in range(nc):
    c = rng.random(3) * 165.0
    s.translate(s.rotate_y(s.sphere(tuple(c), 10.0, white, "cloud"), 15.0), (-100, 270, 395))
s.sphere((260, 150, 45), 50.0, s.material(f.MAT_DIELECTRIC, p=(1.5,)), "glass")
s.sphere((0, 150, 145), 50.0, s.material(f.MAT_METAL, color=(0.8, 0.8, 0.9), p=(1.0,)), "metal")
s.set_sky(1)
s.set_camera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40.0, 1.0)
s.finish()
print(f"boxes {nb}x{nb}, cloud {nc}: {s.flat.n_spheres} spheres, {s.flat.n_rects} rects, {s.flat.n_xforms} wrappers")
rends = []
for label, opts in (("default placement", {}), ("tree through L2 (tree_placement = 1)", {"tree_placement": 1}), ("default again", {})):
    r = rt.Renderer(0)
    for k, v in opts.items():
        r.set_option(k, v)
    r.upload(s)
    rends.append((label, r))
p = rt.make_params(1080, 1080, spp, max_depth=50, flags=_ffi.FLAG_TIME_DEPTHS)
res = {i: [] for i in range(len(rends))}
ref = None
for it in range(rounds + 1):
    for i, (label, r) in enumerate(rends):
        img, _, st = r.render(s.camera, p)
        if ref is None:
            ref = img.copy()
        assert np.array_equal(ref.view(np.uint32), img.view(np.uint32))
        a, b, n = r.depth_timings()
        if it:
            res[i].append((a.sum(), b.sum(), st.seconds_device * 1e3, st.n_rays))
for i, (label, r) in enumerate(rends):
    t = statistics.median(x[2] for x in res[i])
    print(f"{label:50s} isect {statistics.median(x[0] for x in res[i]):8.3f} ms  shade {statistics.median(x[1] for x in res[i]):8.3f} ms  device {t:8.3f} ms"
          f"  {res[i][0][3] / t / 1e3:8.0f} Mray/s")
