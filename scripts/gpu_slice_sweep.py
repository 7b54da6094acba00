"""Frame time against slice size (working set per slice), one library: python scripts/gpu_slice_sweep.py [spp] [slice ...]
Prints the two-chain frame (device ms) and the one-chain per-kernel sums for every slice size; 0 = the default (one slice if it fits)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
slices = [int(x) for x in sys.argv[2:]] or [0, 64, 16, 4]
rt.register_default_images()
scene = rt.Scene.build(os.environ.get("RTOW_SCENE", "sphere_scene"), 16 / 9)
r = rt.Renderer(0)
r.upload(scene)
for s in slices:
    for flags, label in ((0, "two chains"), (rt._ffi.FLAG_TIME_DEPTHS, "one chain ")):
        p = rt.make_params(1920, 1080, spp, max_depth=50, spp_slice=s, flags=flags)
        r.render(scene.camera, p)
        best = None
        for _ in range(3):
            _, _, st = r.render(scene.camera, p)
            a, b, n = r.depth_timings()
            row = (st.seconds_device * 1e3, st.seconds_trace * 1e3, a.sum(), b.sum(), st.n_slices)
            best = row if best is None or row[0] < best[0] else best
        print(f"spp_slice {s:4d} ({best[4]:3d} slices) {label}: device {best[0]:7.2f} ms  trace {best[1]:7.2f} ms"
              + (f"  first slice: isect {best[2]:6.2f}  shade {best[3]:6.2f}" if flags else ""), flush=True)
