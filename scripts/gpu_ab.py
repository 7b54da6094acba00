"""Interleaved A/B of library builds in ONE process on ONE device (cdna guide rule 24):
python scripts/gpu_ab.py <spp> <rounds> libA.so libB.so[@option=value,...] ...   -> median isect/shade/total ms per variant
(options: the names of rt_debug_set_option, _ffi.OPT_NAMES, set on that variant's context before its scene is uploaded;
 "-" as the library = the in-tree librtow_mi355x.so)
(RTOW_SCENE=name picks the scene, RTOW_AB_DEPTHS=N adds the per-depth split of the first N depths)

Since round 5 the order alternates (A B C .. / .. C B A), every row reports median, range and spread of its samples, and the shader
clock and socket power read right after each frame are listed, so a slow row can be told from a slow clock.

Position bias: the variant listed FIRST has been seen to read up to 0.5 ms (2-3 %) high on k_shade with two identical
binaries (gpurun_out/r2/ab_rcp.txt).  List the baseline first AND last, or a copy of the candidate twice, and believe a
difference only when it exceeds the spread between the identical copies.

Noise floor per scene: on cornell_box the SAME binary read k_shade 24.77 ms first and 25.80 ms last in one run (4 %,
gpurun_out/r2/ab_align.txt), and two builds with identical k_shade source differed by as much; on sphere_scene identical
binaries agree to ~1 %.  Judge a k_shade difference on cornell_box or final_scene only against that spread.  A same-binary
switch (an rt_debug_set_option of one context, alternated several times) is the better experiment whenever one is possible.  (-falign-loops=64 / 128: no effect on any scene.)"""
import ctypes
import json
import os
import statistics
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_in_one_weekend_amd as rt
from ray_tracing_in_one_weekend_amd import _ffi


def gpu_state():
    """(shader clock MHz, socket power W) from rocm-smi, or (None, None): logged beside every measurement so that a slow row can be
    told from a slow clock (the chip lowers its clock under load, and boxes differ: MI355X_MICROARCH.md 'DVFS give-back')."""
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        clk = next((v for k, v in card.items() if k.startswith("sclk clock speed")), None)
        pw = next((v for k, v in card.items() if "Socket Graphics Package Power" in k or "Average Graphics Package Power" in k), None)
        return (int(str(clk).strip("()").lower().replace("mhz", "")) if clk else None), (float(pw) if pw else None)
    except Exception:  # noqa: BLE001 — the log column is optional
        return None, None


spp, rounds = int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:]
rt.register_default_images()
scene = rt.Scene.build(os.environ.get("RTOW_SCENE", "sphere_scene"), 16 / 9)
rends = []
default_lib = _ffi.GPU_LIB_PATH
for k, spec in enumerate(libs):  # "lib.so" or "lib.so@option=value[,option2=value2]"
    path, _, opts = spec.partition("@")
    path = default_lib if path == "-" else path
    libs[k] = os.path.basename(path) + ("@" + opts if opts else "")
    _ffi._gpu_lib = None
    _ffi.GPU_LIB_PATH = path
    r = rt.Renderer(0)
    for name, val in (e.split("=", 1) for e in opts.split(",") if e):
        r.set_option(name, int(val))
    r.upload(scene)
    rends.append(r)
p = rt.make_params(1920, 1080, spp, max_depth=50, flags=int(os.environ.get("RTOW_AB_FLAGS", rt._ffi.FLAG_TIME_DEPTHS)))  # 0: the production two-chain frame (device ms only)
res = {i: [] for i in range(len(libs))}
per_depth = {i: [] for i in range(len(libs))}
log = []
for it in range(rounds + 1):
    # A B C ... then ... C B A: every variant sits early and late equally often (a drift over the run — clock, temperature — and
    # any advantage of a position cancel in the median)
    order = list(range(len(rends))) if it % 2 == 0 else list(range(len(rends) - 1, -1, -1))
    for i in order:
        r = rends[i]
        img, _, st = r.render(scene.camera, p)
        a, b, n = r.depth_timings()
        clk, pw = gpu_state() if os.environ.get("RTOW_AB_SMI", "1") != "0" else (None, None)
        if it:  # first round = warm-up
            res[i].append((a.sum(), b.sum(), st.seconds_device * 1e3, st.n_rays, st.n_slices))
            per_depth[i].append((a.copy(), b.copy()))
            log.append((it, i, st.seconds_device * 1e3, clk, pw))
for i, path in enumerate(libs):
    a = statistics.median(x[0] for x in res[i]); b = statistics.median(x[1] for x in res[i])
    ts = sorted(x[2] for x in res[i]); t = statistics.median(ts)
    print(f"{path:40s} isect {a:7.2f} ms  shade {b:7.2f} ms  device {t:7.2f} ms (median of {len(ts)}; {ts[0]:.2f} .. {ts[-1]:.2f}, "
          f"spread {100.0 * (ts[-1] - ts[0]) / t:.1f} %)  rays {res[i][0][3]}  slices {res[i][0][4]}  -> {res[i][0][3] / t / 1e3:8.0f} Mray/s")
print("round variant  device ms   sclk MHz  power W   (the clock right after the frame, rocm-smi)")
for it, i, ms, clk, pw in log:
    print(f"{it:5d} {i:7d}  {ms:9.2f}  {str(clk):>9s}  {str(pw):>7s}")
# RTOW_AB_DEPTHS=N: the first N depths (median ms of k_intersect / k_shade per variant) and the rest as one line
nd = int(os.environ.get("RTOW_AB_DEPTHS", "0"))
if nd:
    import numpy as np
    med = [(np.median([x[0] for x in per_depth[i]], axis=0), np.median([x[1] for x in per_depth[i]], axis=0)) for i in range(len(libs))]
    print("depth  " + "  ".join(f"{os.path.basename(p)[:18]:>18s} isect/shade" for p in libs))
    for d in range(min(nd, len(med[0][0]))):
        print(f"{d:5d}  " + "  ".join(f"{m[0][d]:18.3f} /{m[1][d]:8.3f}  " for m in med))
    print(" rest  " + "  ".join(f"{m[0][nd:].sum():18.3f} /{m[1][nd:].sum():8.3f}  " for m in med))
