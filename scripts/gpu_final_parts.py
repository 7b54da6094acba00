"""Where final_scene's k_intersect time goes: a look-alike built through the piecewise API with parts left out
(diagnostic): python scripts/gpu_final_parts.py [spp]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_in_one_weekend_amd as rt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rt.register_default_images()
f = rt._ffi


def build(cloud, boxes, wrap_cloud=True):
    rng = np.random.default_rng(7)
    s = rt.Scene.new()
    ground = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.48, 0.83, 0.53)))
    white = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.73, 0.73, 0.73)))
    light = s.material(f.MAT_EMISSION, tex0=s.constant_tex((7, 7, 7)))
    glass = s.material(f.MAT_DIELECTRIC, p=(1.5, 0, 0, 0))
    s.rect(f.RECT_XZ, (123, 544, 147), (423, 554, 412), light)
    if boxes:
        for i in range(20):
            for j in range(20):
                x0, z0 = -1000.0 + i * 100.0, -1000.0 + j * 100.0
                s.gbox((x0, 0.0, z0), (x0 + 100.0, float(rng.uniform(1, 101)), z0 + 100.0), ground)
    else:
        s.rect(f.RECT_XZ, (-1000, 50, -1000), (1000, 50, 1000), ground)
    if cloud:
        for _ in range(1000):
            c = rng.uniform(0, 165, 3)
            if wrap_cloud:
                s.translate(s.rotate_y(s.sphere(tuple(c), 10.0, white, "c"), 15.0), (-100, 270, 395))
            else:
                s.sphere((float(c[0]) - 100, float(c[1]) + 270, float(c[2]) + 395), 10.0, white, "c")
    s.sphere((400, 200, 400), 100.0, white, "a")
    s.sphere((260, 150, 45), 50.0, glass, "g")
    s.constant_medium(s.sphere((360, 150, 145), 50.0, glass, "b"), 0.2, s.constant_tex((0.2, 0.4, 0.9)))
    s.set_sky(f.SKY_BLACK)
    s.set_camera((478, 278, -600), (278, 278, 0), (0, 1, 0), 40, 1.0)
    s.finish()
    return s


r = rt.Renderer(0)
p = rt.make_params(1080, 1080, spp, max_depth=50, flags=f.FLAG_TIME_DEPTHS)
for name, sc in (("boxes + cloud", build(True, True)), ("boxes + cloud, un-instanced", build(True, True, False)),
                 ("boxes only", build(False, True)), ("cloud only (flat floor)", build(True, False)),
                 ("neither", build(False, False))):
    r.upload(sc)
    r.render(sc.camera, p)
    res = []
    for _ in range(3):
        _, _, st = r.render(sc.camera, p)
        a, b, n = r.depth_timings()
        res.append((a.sum(), b.sum(), st.n_rays))
    a = statistics.median(x[0] for x in res)
    b = statistics.median(x[1] for x in res)
    print(f"{name:30s} prims {sc.flat.n_spheres + sc.flat.n_rects:5d}  isect {a:7.2f} ms  shade {b:6.2f} ms  rays {res[0][2]}  isect {res[0][2] / a / 1e6:6.2f} Gray/s", flush=True)
