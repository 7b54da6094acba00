"""Throughput of every mirrored reference scene on one GPU, with the CPU oracle timed beside it:
python scripts/gpu_scene_table.py [spp] [out.json]
The GPU renders the full frame (1920x1080 for the 16:9 scenes, 1080x1080 for the two square ones); the oracle
renders a bounded crop-free sample (smaller frame, fewer samples) on all usable cores."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt
from oracle import binding as orc

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out_path = sys.argv[2] if len(sys.argv) > 2 else None
rt.register_default_images()
orc.load()
rend = rt.Renderer(0)
rows = []
for name, aspect, nx, ny in [("sphere_scene", 16 / 9, 1920, 1080), ("test_sphere", 16 / 9, 1920, 1080),
                             ("simple_light_scene", 16 / 9, 1920, 1080), ("cornell_box", 1.0, 1080, 1080),
                             ("final_scene", 1.0, 1080, 1080), ("earth_env_scene", 16 / 9, 1920, 1080),
                             ("pbr_sweep_scene", 16 / 9, 1920, 1080)]:
    scene = rt.Scene.build(name, aspect)
    rend.upload(scene)
    p = rt.make_params(nx, ny, spp, max_depth=50)
    rend.render(scene.camera, p)  # warm-up
    runs = [rend.render(scene.camera, p)[2] for _ in range(3)]
    dev = statistics.median(s.seconds_device for s in runs)
    n_rays = int(runs[0].n_rays)
    cp = rt.make_params(nx // 8, ny // 8, 4, max_depth=50)
    t0 = time.perf_counter()
    _, _, so = orc.render(scene.flat_ptr, scene.camera, cp, orc.options(rng_mode=orc.RNG_STREAM, accel=orc.ACCEL_BVH))
    cpu_s = time.perf_counter() - t0
    row = {"scene": name, "frame": f"{nx}x{ny}x{spp}spp", "n_prims": int(scene.flat.n_spheres + scene.flat.n_rects),
           "n_media": int(scene.flat.n_media), "gpu_ms": dev * 1e3, "gpu_mray_s": n_rays / dev / 1e6,
           "rays_per_path": n_rays / (nx * ny * spp), "cpu_mray_s": int(so.n_rays) / cpu_s / 1e6,
           "cpu_sample": f"{nx // 8}x{ny // 8}x4spp stream-mode BVH oracle, all cores"}
    rows.append(row)
    print(f"{name:20s} prims {row['n_prims']:5d}  {row['gpu_ms']:8.2f} ms  {row['gpu_mray_s']:9.0f} Mray/s  "
          f"{row['rays_per_path']:5.2f} rays/path   cpu {row['cpu_mray_s']:6.1f} Mray/s", flush=True)
if out_path:
    with open(out_path, "w") as f:
        json.dump(rows, f, indent=1)
