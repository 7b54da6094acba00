"""Per-depth kernel times of one slice (diagnostic): python scripts/gpu_depth_probe.py [spp] [flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rt.register_default_images()
scene = rt.Scene.build(sys.argv[3] if len(sys.argv) > 3 else "sphere_scene", 16 / 9)
r = rt.Renderer(0)
r.upload(scene)
p = rt.make_params(1920, 1080, spp, max_depth=50, flags=flags | rt._ffi.FLAG_TIME_DEPTHS)
for it in range(2):
    img, _, st = r.render(scene.camera, p)
a, b, n = r.depth_timings()
print(f"rays {st.n_rays} device {st.seconds_device*1e3:.2f} ms trace {st.seconds_trace*1e3:.2f} ms -> {st.n_rays/st.seconds_device/1e6:.0f} Mray/s")
print("depth      rays   isect_us  shade_us   isect_Gray/s shade_Gray/s")
for d in range(len(a)):
    if d < 14 or d % 6 == 0:
        gi = n[d] / max(a[d], 1e-6) / 1e6
        gs = n[d] / max(b[d], 1e-6) / 1e6 if b[d] > 0 else 0
        print(f"{d:3d} {int(n[d]):10d} {a[d]*1e3:9.1f} {b[d]*1e3:9.1f} {gi:10.2f} {gs:10.2f}")
print("sum isect %.2f ms shade %.2f ms" % (a.sum(), b.sum()))
