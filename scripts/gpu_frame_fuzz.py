"""Fuzz of what a frame must not depend on, on the GPU box: python scripts/gpu_frame_fuzz.py [n] [first_seed]
Every seed draws a scene (a random general scene of tests/test_gpu_parity.py or a sphere-only scene of scripts/gpu_grid_fuzz.py), a
frame size from 1 x 1 to 210 x 150 (sizes that are no multiple of the 8 x 8 slot tiles, the 64-lane waves or the band), 1-9
samples, depth 1-20, and renders it (a) whole, (b) in row-interleaved shards of a random band and count, each in random sample
slices, (c) with the primary candidate lists off, (d) with the other closest-hit structure (grid off), (e) tiles of the other
pixel order — all five must be the same bits with the same ray count per depth — and (f) by the list walk, which may differ only
in rays whose fp32 Sphere::hit root is a miss in exact arithmetic (sphere scenes: such a frame is re-traced and every ray at
which list walk and structures part is checked in float64; general scenes: counted and reported).  Device against device: the oracle is not involved."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracing_in_one_weekend_amd as rt  # noqa: E402
from ray_tracing_in_one_weekend_amd import shard  # noqa: E402
from helpers import grid_fuzz_scene  # noqa: E402
import test_gpu_parity as T  # noqa: E402



def explain_list_walk(scene, nx, ny, spp, depth, pseed):
    """A frame whose list walk differs from the structures: every path re-traced along the list walk's own results (rt_debug_bounce's
    key-honouring kernels), each ray also put through the production closest-hit kernels; where the two part, the spheres involved
    in float64: (closest approach / r)^2 > 1 means the exact ray misses the sphere — the list walk's root is an fp32 false positive
    that a box or cell test culled, anything else is a bug.  Returns the number of unexplained rays."""
    f = rt._ffi
    q = rt.make_params(nx, ny, spp, max_depth=depth, seed=pseed)
    a = scene.arrays()
    sc_ = np.stack([a["sph_cx"], a["sph_cy"], a["sph_cz"]], 1).astype(np.float64)
    sr_ = np.abs(a["sph_r"].astype(np.float64))
    pj, pi = np.divmod(np.arange(nx * ny), nx)
    o, d, keys = T._primary_rays(scene, q, np.repeat(pi, spp), np.repeat(pj, spp), np.tile(np.arange(spp), nx * ny))
    live = np.ones(len(o), dtype=bool)
    bad = 0
    for dep in range(depth + 1):
        idx = np.flatnonzero(live)
        if not len(idx):
            break
        g = r.debug_bounce(o[idx], d[idx], keys[idx], depth=dep, flags=f.FLAG_BRUTE_FORCE)
        for grid_off in (0, 1):  # the grid where the scene has one, and the tree (which also answers depth 0 of the overflowing pixels)
            r.set_option("grid", grid_off)
            gp_ = r.debug_bounce(o[idx], d[idx], keys[idx], depth=dep, flags=f.FLAG_PRODUCTION_KERNELS)
            r.set_option("grid", 0)
            for k in np.flatnonzero((gp_["hit"] != g["hit"]) | (gp_["t"].view(u) != g["t"].view(u))):
                oo, dd = o[idx[k]].astype(np.float64), d[idx[k]].astype(np.float64)
                h = int(g["hit"][k])
                oc = oo - sc_[h] if h >= 0 else None
                ratio = (oc @ oc - (oc @ dd) ** 2 / (dd @ dd)) / sr_[h] ** 2 if h >= 0 else 0.0
                ok = h >= 0 and ratio > 1.0 and (gp_["hit"][k] < 0 or gp_["t"][k] >= g["t"][k])
                bad += 0 if ok else 1
                print(f"    depth {dep} ({'tree' if grid_off else 'grid'}): list walk hit {h} t {float(g['t'][k])!r}, structures hit {int(gp_['hit'][k])} t {float(gp_['t'][k])!r}; "
                      f"(closest approach / r)^2 of sphere {h} = {ratio:.9f}" + ("  (fp32 false positive of Sphere::hit)" if ok else "  <-- UNEXPLAINED"))
        alive = g["alive"].astype(bool)
        o[idx[alive]], d[idx[alive]] = g["o"][alive], g["d"][alive]
        live[idx[~alive]] = False
    return bad


n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
far_exp = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0  # general scenes are moved up to 10^far_exp units out
rt.register_default_images()
r = rt.Renderer(0)
u = np.uint32
n_list_differs = 0
t0 = time.time()
for seed in range(first, first + n_seeds):
    rng = np.random.default_rng(10_000 + seed)
    kind = "general" if rng.random() < 0.5 else "spheres"
    far = kind == "general" and rng.random() < 0.5  # the general scene moved 1e3 .. 1e6 units out as a whole
    off = rng.normal(size=3) * 10.0 ** rng.uniform(3, far_exp) if far else None
    scene = (T._random_scene(rt, 1000 + int(rng.integers(0, 100000)), offset=off) if kind == "general"
             else grid_fuzz_scene(rt, int(rng.integers(0, 100000)))[0])
    if far:
        kind = f"general at {np.abs(off).max():.0e}"
    nx = int(rng.choice([1, 2, 7, 8, 9, 31, 33, 63, 64, 65, 100, 129, 210]))
    ny = int(rng.choice([1, 2, 3, 7, 8, 9, 17, 50, 64, 65, 150]))
    spp = int(rng.integers(1, 10))
    depth = int(rng.integers(1, 21))
    pseed = int(rng.integers(0, 2**40))
    r.upload(scene)

    def frame(**kw):
        img, _, st = r.render(scene.camera, rt.make_params(nx, ny, spp, max_depth=depth, seed=pseed, **kw))
        return img, list(st.rays_per_depth)[:depth + 2]

    whole, rays = frame()
    what = []
    # (b) shards x slices
    world = int(rng.integers(2, 6))
    band = int(rng.choice([1, 2, 3, 8, 16]))
    parts, rsum = [], np.zeros(len(rays), dtype=np.int64)
    for k in range(world):
        img, rr = frame(shard_band=band, shard_count=world, shard_id=k, spp_slice=int(rng.integers(1, spp + 1)))
        parts.append(img)
        rsum += np.array(rr)
    sharded = shard.deinterleave(parts, ny, band, world)
    if not (np.array_equal(sharded.view(u), whole.view(u)) and rsum.tolist() == rays):
        what.append(f"shards (band {band} x {world})")
    # (c) lists off, (d) the other structure, (e) the other pixel order
    for opt, val, name in (("primary_lists", 1, "lists off"), ("grid", 1, "grid off"), ("pixel_order", 1, "pixel order 1"), ("pixel_order", 2, "pixel order 2")):
        r.set_option(opt, val)
        try:
            img, rr = frame()
        finally:
            r.set_option(opt, 0)
        if not (np.array_equal(img.view(u), whole.view(u)) and rr == rays):
            what.append(name)
    img, rr = frame(flags=rt._ffi.FLAG_BRUTE_FORCE)
    lw = np.array_equal(img.view(u), whole.view(u)) and rr == rays
    n_list_differs += 0 if lw else 1
    if not lw and kind == "spheres" and explain_list_walk(scene, nx, ny, spp, depth, pseed):
        what.append("list walk, unexplained")
    print(f"seed {seed}: {kind}, {nx} x {ny} x {spp} spp, depth {depth}, {sum(rays)} rays: "
          + ("ok" if not what else "DIFFERS: " + ", ".join(what)) + ("" if lw else "  (list walk differs)") + f"  {time.time() - t0:6.1f} s", flush=True)
    if what:
        sys.exit(1)
print(f"{n_seeds} frames, seeds {first}..{first + n_seeds - 1}: whole == shards x slices == lists off == other structure == other pixel orders; "
      f"the list walk differs in {n_list_differs}")
r.close()
