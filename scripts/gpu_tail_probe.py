"""The deep end of the bounce loop with and without k_tail (diagnostic): python scripts/gpu_tail_probe.py <tail_depth> [frames]
Renders config 2 `frames` times with RT_OPT_TAIL_DEPTH = <tail_depth> (1 = never); run it under
`rocprofv3 --kernel-trace --stats` to see what the launches past that depth cost as dispatches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

rt.register_default_images()
scene = rt.Scene.build("sphere_scene", 16 / 9)
r = rt.Renderer(0)
r.set_option("tail_depth", int(sys.argv[1]))
if len(sys.argv) > 3:
    r.set_option("chains", int(sys.argv[3]))
r.upload(scene)
p = rt.make_params(1920, 1080, 256, max_depth=50, seed=95)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    _, _, st = r.render(scene.camera, p)
    print(f"tail_depth {sys.argv[1]}: device {st.seconds_device * 1e3:.2f} ms, {st.n_trace_launches} launches", flush=True)
