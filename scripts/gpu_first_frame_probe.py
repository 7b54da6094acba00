"""Why is the first frame of a fresh process slower on the device than the second?  Renders, in fresh contexts of one process:
(a) config 2 twice; (b) a tiny frame first (code loaded, queues mapped, clocks up), then config 2 twice with freshly allocated buffers;
(c) config 2, close the context (buffers freed), a new context, config 2 again.  Device ms of every frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

rt.register_default_images()
scene = rt.Scene.build("sphere_scene", 16 / 9)
big = rt.make_params(1920, 1080, 256, max_depth=50, seed=95)
tiny = rt.make_params(64, 36, 4, max_depth=50, seed=95)


def frames(r, params, n):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        _, _, st = r.render(scene.camera, params)
        out.append((round(st.seconds_device * 1e3, 2), round((time.perf_counter() - t0) * 1e3, 2)))
    return out


r = rt.Renderer(0)
r.upload(scene)
print("(a) fresh process, config 2 x3 (device ms, wall ms):", frames(r, big, 3), flush=True)
r.close()
r = rt.Renderer(0)
r.upload(scene)
print("(c) new context after the first was closed, config 2 x3:", frames(r, big, 3), flush=True)
r.close()
time.sleep(3.0)
r = rt.Renderer(0)
r.upload(scene)
print("(d) new context after 3 s of idle, config 2 x3:", frames(r, big, 3), flush=True)
r.close()
time.sleep(3.0)
r = rt.Renderer(0)
r.upload(scene)
print("(b) after 3 s of idle: tiny frame x2 then config 2 x3:", frames(r, tiny, 2), frames(r, big, 3), flush=True)
r.close()
