"""More seeds of tests/test_gpu_parity.py::test_random_scenes_bounce_parity than the suite's 48 (a checker run on the GPU box: the
oracle is the CPU side of every comparison): python scripts/gpu_random_scene_sweep.py [first_seed] [n]
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
rt.register_default_images()
orc.load()
r = rt.Renderer(0)
fn = getattr(T.test_random_scenes_bounce_parity, "__wrapped__", T.test_random_scenes_bounce_parity)
t0 = time.time()
for seed in range(first, first + n):
    fn(rt, orc, r, seed)
    s = T._random_scene(rt, 1000 + seed).flat
    print(f"seed {seed}: ok  ({s.n_spheres} spheres, {s.n_rects} rectangles, {s.n_xforms} wrappers, {s.n_media} media, sky {s.sky_type})  {time.time() - t0:6.1f} s", flush=True)
print(f"{n} scenes, seeds {first}..{first + n - 1}: all per-ray records and frames agree with the oracle")
r.close()
