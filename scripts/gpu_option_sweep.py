"""Interleaved sweep of one rt_debug_set_option in one process:
    python scripts/gpu_option_sweep.py <scene> <spp> <rounds> <option> v0 v1 ...   -> median isect / shade / device ms per value
`option` is a name of _ffi.OPT_NAMES (tree_placement, primary_lists, pixel_order, texel_pool, grid, grid_cell, chains, general_kernels,
general_lds, queue_shards, isect_workgroups, materialise_primaries); every value gets a context of its own (upload-time options
take effect there), so at most ~8 values of a 128-spp frame fit the device.  RTOW_SWEEP_FLAGS=0 renders the production frame (two
chains, no per-depth timing) instead of the single chain that per-depth timing needs."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

scene_name, spp, rounds, option = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
values = [int(v) for v in sys.argv[5:]]
rt.register_default_images()
square = scene_name in ("cornell_box", "final_scene")
scene = rt.Scene.build(scene_name, 1.0 if square else 16 / 9)
rends = []
for v in values:
    r = rt.Renderer(0)
    r.set_option(option, v)
    r.upload(scene)
    rends.append(r)
flags = int(os.environ.get("RTOW_SWEEP_FLAGS", rt._ffi.FLAG_TIME_DEPTHS))
p = rt.make_params(1080 if square else 1920, 1080, spp, max_depth=50, flags=flags)
res = {i: [] for i in range(len(values))}
ref = None
for it in range(rounds + 1):
    for i, r in enumerate(rends):
        img, _, st = r.render(scene.camera, p)
        if ref is None:
            ref = img.copy()
        assert (ref.view("uint32") == img.view("uint32")).all(), (option, values[i])  # every setting renders the same bits
        a, b, _ = r.depth_timings() if flags else ([0.0], [0.0], None)
        if it:
            res[i].append((float(sum(a)), float(sum(b)), st.seconds_device * 1e3, st.n_rays))
for i, v in enumerate(values):
    a, b, t = (statistics.median(x[k] for x in res[i]) for k in range(3))
    print(f"{option} = {v:<8d} isect {a:7.2f} ms  shade {b:7.2f} ms  device {t:7.2f} ms  -> {res[i][0][3] / t / 1e3:8.0f} Mray/s   {rends[i].scene_info()}")
