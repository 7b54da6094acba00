"""Finds where a GPU frame and the oracle's part: python scripts/gpu_debug_divergence.py <scene> <nx> <ny> <spp> <max_depth>
Renders both, takes the pixels that differ, re-traces every sample of those pixels bounce by bounce through rt_debug_bounce on
both sides and prints the first bounce at which the two disagree."""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import ray_tracing_in_one_weekend_amd as rt
from oracle import binding as orc
from test_gpu_parity import _primary_rays

name, nx, ny, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rt.register_default_images()
orc.load()
scene = rt.Scene.build(name, nx / ny)
r = rt.Renderer(0)
r.upload(scene)
p = rt.make_params(nx, ny, spp, max_depth=depth)
img, _, st = r.render(scene.camera, p)
it, _, so = orc.render(scene.flat_ptr, scene.camera, p, orc.options(rng_mode=orc.RNG_COUNTER, estimator=orc.EST_ITERATIVE))
print("rays", st.n_rays, so.n_rays, [a - b for a, b in zip(list(st.rays_per_depth)[:depth + 1], list(so.rays_per_depth)[:depth + 1])])
diff = np.abs(img.astype(np.float64) - it).max(axis=2)
jj, ii = np.nonzero(diff > 1e-5)
print(len(jj), "pixels differ by more than 1e-5 (linear)")
for pj, pi in list(zip(jj, ii))[:6]:
    n = spp
    o, d, keys = _primary_rays(scene, p, np.full(n, pi), np.full(n, pj), np.arange(n))
    live = np.ones(n, bool)
    for dep in range(depth + 1):
        idx = np.nonzero(live)[0]
        if not len(idx):
            break
        g = r.debug_bounce(o[idx], d[idx], keys[idx], depth=dep)
        c = orc.debug_bounce(scene.flat_ptr, o[idx], d[idx], keys[idx], depth=dep, accel=orc.ACCEL_LIST)
        bad = [k for k in range(len(idx)) if g["hit"][k] != c["hit"][k] or g["alive"][k] != c["alive"][k] or
               not np.array_equal(g["o"][k].view(np.uint32), c["o"][k].view(np.uint32)) or not np.array_equal(g["d"][k].view(np.uint32), c["d"][k].view(np.uint32))]
        for k in bad[:3]:
            print(f"pixel ({pi},{pj}) sample {idx[k]} depth {dep}: hit {g['hit'][k]} / {c['hit'][k]}  t {g['t'][k]!r} / {c['t'][k]!r}  alive {g['alive'][k]} / {c['alive'][k]}")
            print("   in  o", o[idx[k]], "d", d[idx[k]])
            print("   gpu o", g["o"][k], "d", g["d"][k], "att", g["attenuation"][k])
            print("   orc o", c["o"][k], "d", c["d"][k], "att", c["attenuation"][k])
        if bad:
            break
        alive = g["alive"].astype(bool)
        o[idx[alive]], d[idx[alive]] = g["o"][alive], g["d"][alive]
        live[idx[~alive]] = False
