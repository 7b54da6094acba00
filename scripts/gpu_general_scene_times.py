"""Device time of the general scenes (wrappers, rectangles, media) with the library of THIS tree, GPU only:
python scripts/gpu_general_scene_times.py [spp] [repeats] [scene ...]
Prints one line per scene: median / min device ms of a 1080 x 1080 frame.  For an A/B of two trees on one box run it from each tree in
turn (a tree's own package is the one imported: sys.path[0] is the directory above this script)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
names = sys.argv[3:] or ["cornell_box", "final_scene", "simple_light_scene"]
rt.register_default_images()
rend = rt.Renderer(0)
print(f"library {rend.build_id}  ({os.path.dirname(os.path.abspath(rt.__file__))})")
for name in names:
    scene = rt.Scene.build(name, 1.0)
    rend.upload(scene)
    p = rt.make_params(1080, 1080, spp, max_depth=50, flags=rt._ffi.FLAG_TIME_DEPTHS)  # (general scenes run one chain anyway)
    rend.render(scene.camera, p)
    runs, isect, shade = [], [], []
    for _ in range(reps):
        runs.append(rend.render(scene.camera, p)[2])
        a, b, _n = rend.depth_timings()
        isect.append(float(a.sum())), shade.append(float(b.sum()))
    ms = sorted(s.seconds_device * 1e3 for s in runs)
    print(f"{name:20s} {statistics.median(ms):8.2f} ms median  {ms[0]:8.2f} min  {ms[-1]:8.2f} max  closest hit {statistics.median(isect):7.2f}  shading {statistics.median(shade):7.2f}  "
          f"{int(runs[0].n_rays) / statistics.median(ms) / 1e3:8.0f} Mray/s", flush=True)
