"""More seeds of tests/test_gpu_parity.py::test_random_scenes_bounce_parity than the suite's 48 (a checker run on the GPU box: the
oracle is the CPU side of every comparison): python scripts/gpu_random_scene_sweep.py [first_seed] [n] [far_exponent] [nesting]
(nesting = 1: 3-9 wrappers around every object, 33-47 media, wrappers around media — what tests/test_gpu_parity.py
test_wrappers_and_media_nest_like_the_trait_objects holds for eight seeds)
(far_exponent e: every other scene is moved as a whole — one more Translate around every object, camera and rays with it — 10^3 ..
10^e units from the origin: fp32 geometry far out, and Perlin lattice indices beyond 2^31 from 3.4e7 on)
Every seed builds a random general scene (spheres, rectangles, boxes, wrappers, media, every material and texture kind), holds
30 000 rays per ray against the list walk on the device and against the oracle, and a small frame through the whole pipeline
against the oracle with every outlier re-traced.  Prints one line per seed and stops at the first failure."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracing_in_one_weekend_amd as rt  # noqa: E402
from oracle import binding as orc  # noqa: E402  (the checker)
import test_gpu_parity as T  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
far_exp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
nesting = len(sys.argv) > 4 and sys.argv[4] == "1"
import numpy as np  # noqa: E402
rt.register_default_images()
orc.load()
r = rt.Renderer(0)
fn = getattr(T.test_random_scenes_bounce_parity, "__wrapped__", T.test_random_scenes_bounce_parity)
t0 = time.time()
n_far = 0
for seed in range(first, first + n):
    off = None
    if far_exp > 3.0 and seed % 2:
        orng = np.random.default_rng(77_000 + seed)
        off = orng.normal(size=3) * 10.0 ** orng.uniform(3.0, far_exp)
        n_far += 1
    fn(rt, orc, r, seed, offset=off, nesting=nesting)
    scene = T._random_scene(rt, 1000 + seed, nesting=nesting)  # (kept alive: .flat points into it)
    s = scene.flat
    print(f"seed {seed}: ok  " + ("" if off is None else f"[moved {np.abs(off).max():.1e} out] ") + f"({s.n_spheres} spheres, {s.n_rects} rectangles, {s.n_xforms} wrappers, {s.n_media} media, sky {s.sky_type})  {time.time() - t0:6.1f} s", flush=True)
print(f"{n} scenes{' with deep nesting' if nesting else ''}, seeds {first}..{first + n - 1} ({n_far} of them moved up to 1e{far_exp:g} units out): all per-ray records and frames agree with the oracle")
r.close()
