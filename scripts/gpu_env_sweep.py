"""Interleaved sweep of an environment knob in one process:
python scripts/gpu_env_sweep.py <scene> <spp> <rounds> <ENV_NAME> v0 v1 ...   -> median device ms per value
Every value gets its own context (knobs read by rt_scene_upload) and the variable is also set around each render
(knobs read by rt_render); the value "-" means unset.  RTOW_SWEEP_FLAGS=0 renders the production frame (two chains on two
streams, no per-depth timing) instead of the single chain that per-depth timing needs."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt

scene_name, spp, rounds, env = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
values = sys.argv[5:]
rt.register_default_images()
square = scene_name in ("cornell_box", "final_scene")
scene = rt.Scene.build(scene_name, 1.0 if square else 16 / 9)
rends = []
_last = []


def setenv(v):
    """`env` = "+": every value is a list NAME=VALUE[,NAME=VALUE...] (several knobs at once; "-" = none set)"""
    if env == "+":
        for k in _last:
            os.environ.pop(k, None)
        _last.clear()
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("=", 1)
                os.environ[k] = val
                _last.append(k)
        return
    if v == "-":
        os.environ.pop(env, None)
    else:
        os.environ[env] = v


one_ctx = os.environ.get("RTOW_SWEEP_ONE_CTX") == "1"  # a knob read per render: one context for all values (its buffers, its slice size)
for v in values:
    setenv(v)
    r = rends[0] if (one_ctx and rends) else rt.Renderer(0)
    if not (one_ctx and rends):
        r.upload(scene)
    rends.append(r)
os.environ.pop(env, None)
p = rt.make_params(1080 if square else 1920, 1080, spp, max_depth=50, flags=int(os.environ.get("RTOW_SWEEP_FLAGS", rt._ffi.FLAG_TIME_DEPTHS)))  # 0: the production two-chain frame
res = {v: [] for v in values}
ref = None
for it in range(rounds + 1):
    for v, r in zip(values, rends):
        setenv(v)
        img, _, st = r.render(scene.camera, p)
        setenv("-")
        a, b, n = r.depth_timings()
        if ref is None:
            ref = img.copy()
        assert (img.view("uint32") == ref.view("uint32")).all(), f"{env}={v} changes the image"
        if it:
            res[v].append((st.seconds_device * 1e3, st.n_rays, a.sum(), b.sum()))
for v in values:
    t = statistics.median(x[0] for x in res[v])
    print(f"{scene_name:14s} {env}={v:>34s}: {t:8.2f} ms  {res[v][0][1] / t / 1e3:8.0f} Mray/s  isect {statistics.median(x[2] for x in res[v]):7.2f} shade {statistics.median(x[3] for x in res[v]):7.2f}", flush=True)
