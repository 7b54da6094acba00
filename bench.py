#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X wavefront path tracer.

Metric (BASELINE.json): Mray/s (primary + secondary) on demo_scene.rs `sphere_scene`
("random-spheres", 533 spheres) at 1920x1080, 256 spp, max depth 50 — config 2, the default on ONE GPU.
    --config 3   sphere_scene 3840x2160, 1024 spp            (BASELINE.json configs[2]: the frame north_star names for the
                                                              8 GPUs of a node — the default when --gpus N > 1, strong scaling)
    --config 4   earth_env_scene 1920x1080, 512 spp           (configs[3]: ImageTex + environment sky)
    --config 5   pbr_sweep_scene 1920x1080, 4096 spp          (configs[4]: pbr.rs sweep)

A "step" is one full render of the frame: every pixel x every sample through the wavefront kernels
(k_primary_lists, then per depth k_intersect -> k_shade, k_resolve per slice, k_finalize), framebuffer resident in
HBM.  With N > 1 ranks (one process per GPU, torch.distributed, backend nccl == RCCL over xGMI) the image rows are
sharded in interleaved bands of 8 rows and each step ends with the gather of the band buffers to rank 0.
    --scaling weak    (default for config 2) per-GPU work fixed: every rank renders (ny/N rows) x nx x spp*N samples
    --scaling strong  (default for configs 3-5) the stated frame split over the N ranks
With N > 1 and no --config the line is config 3 strong, and a short config-2 weak leg (2 steps) rides along under "also"; each names
the committed one-GPU line of its frame on the same kernels ("single_gpu_reference": efficiency = value / (N x its value));
the line carries the gather time per step ("gather_ms", HIP events around the gather + rank 0's de-interleave) and the trace-step
HBM fraction of every rank ("roofline.per_rank").

`python bench.py --gpus N` without a torchrun environment starts the N rank processes itself (a child
`python -m torch.distributed.run` started BEFORE this process touches a GPU) and relays rank 0's JSON line.

Prints ONE JSON line on rank 0 (see the contract in the task statement) including
  "roofline":     trace-step algorithmic HBM bytes / its device time (HIP events on the launch stream)
  "first_frame":  a fresh process's rt_ctx_create + scene build + rt_scene_upload + first rt_render ("first_frame_ms"), "alloc_bytes"
  "cpu_baseline": the CPU oracle in reference (stream) order, all host cores, bounded sample; built -O3 -march=native on
                  the box it is timed on when g++ is there (BASELINE.md's recipe), else the prebuilt -march=x86-64-v2 library
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak

CONFIGS = {  # BASELINE.json configs[1..4]
    2: dict(scene="sphere_scene", nx=1920, ny=1080, spp=256, scaling="weak",
            what="demo_scene.rs sphere_scene (random-spheres, 533 spheres)"),
    3: dict(scene="sphere_scene", nx=3840, ny=2160, spp=1024, scaling="strong",
            what="demo_scene.rs sphere_scene (random-spheres, 533 spheres)"),
    4: dict(scene="earth_env_scene", nx=1920, ny=1080, spp=512, scaling="strong",
            what="earth_env_scene (earthmap.jpg ImageTex spheres, newport_loft.jpg tex_sky_color)"),
    5: dict(scene="pbr_sweep_scene", nx=1920, ny=1080, spp=4096, scaling="strong",
            what="pbr_sweep_scene (pbr.rs GGX metal / plastic / clearcoat sweep, 501 spheres)"),
}


def _latest_profile(pattern, workload_key, build_id):
    """Newest profiles/round*/<pattern> of this run's workload AND of the kernels that are running: a profile records the
    rt_build_id() of the library it was taken on (hash of the device sources + flags); one taken on other kernels is refused.
    Returns (path, json) or (None, reason)."""
    import glob
    best, stale = None, None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", pattern))):
        try:
            t = json.load(open(p))
        except (OSError, ValueError):
            continue
        if (t.get("bench_config") or {}).get("workload") != workload_key:
            continue
        if t.get("library_build_id") != build_id:
            stale = f"{os.path.relpath(p, ROOT)} was taken on build {t.get('library_build_id')}, this library is {build_id}"
            continue
        best = (p, t)
    return best if best else (None, stale)


def pmc_traffic(workload_key, build_id, bytes_per_launch=None):
    """HBM bytes per trace-step launch from the committed PMC pass of this same command on this same build
    (scripts/collect_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes, corrections calibrated in
    profiles/round3/fetch_calibration.json).  bench.py cannot profile itself, so the number comes from profiles/;
    null (with the reason as the source) when there is none for these kernels."""
    p, t = _latest_profile("traffic*.json", workload_key, build_id)
    if not p:
        return None, t
    # per launch of THIS run: the PMC pass profiles the first frame of a process, which may take its samples in more slices (and
    # launches) than a steady frame does — the ratio to the algorithmic bytes is what carries over
    if bytes_per_launch and t.get("traffic_over_algorithmic"):
        return round(t["traffic_over_algorithmic"] * bytes_per_launch, 1), os.path.relpath(p, ROOT)
    return t["trace_step_bytes_per_launch"], os.path.relpath(p, ROOT)


def pmc_valu(workload_key, build_id):
    """VALU issue fraction / lane utilisation of the trace kernels from the committed PMC pass of this same command on this
    same build (scripts/collect_valu.py) against the MEASURED issue rate of scripts/micro/mul_rate.hip; None when absent."""
    p, t = _latest_profile("valu*.json", workload_key, build_id)
    if not p:
        return None
    out = {k: {"issue_frac": round(v["issue_frac"], 4), "lane_util": round(v["lane_util"], 4)} for k, v in t["kernels"].items()}
    out["peak_wave_insts_per_s"] = t.get("peak_wave_insts_per_s")
    out["source"] = os.path.relpath(p, ROOT)
    out["note"] = ("companion roofline: SQ_INSTS_VALU / (kernel seconds x measured peak issue rate); the peak is the v_mul_f32 / "
                   "v_add_f32 rate of scripts/micro/mul_rate.hip at 8 waves per SIMD (2.1 cycles per wave64 instruction on the "
                   "SIMD-32, profiles/round2/valu_peak.json), so issue_frac <= 1; lane_util = active lanes per issued instruction")
    return out


def single_gpu_reference(config_id, metric, build_id, profiles_root=None):
    """The like-for-like one-GPU figure for a run on N > 1 GPUs: the committed `bench.py --config <id>` line of ONE GPU for the
    same frame (profiles/round*/bench_config<id>.json: same metric label = same nx, ny, spp) taken on THESE kernels — a record of
    another build is refused, as the PMC files are.  With it scaling efficiency = value / (n_gpus x reference value) can be read
    off the one line.  Returns the dict for the "single_gpu_reference" key (value None and a reason when there is no such record)."""
    import glob
    root = profiles_root or os.path.join(ROOT, "profiles")
    reason = f"no profiles/round*/bench_config{config_id}.json"
    best = None
    for p in sorted(glob.glob(os.path.join(root, "round*", f"bench_config{config_id}.json"))):
        try:
            t = json.load(open(p))
        except (OSError, ValueError):
            continue
        if t.get("n_gpus") != 1 or t.get("metric") != metric or not t.get("value"):
            reason = f"{os.path.relpath(p, os.path.dirname(root))} is not a one-GPU line of this frame"
            continue
        if t.get("library_build_id") != build_id:
            reason = f"{os.path.relpath(p, os.path.dirname(root))} was taken on build {t.get('library_build_id')}, this library is {build_id}"
            continue
        best = {"config": config_id, "value": t["value"], "unit": t.get("unit", "Mray/s"), "ms_per_step": t.get("ms_per_step"),
                "source": os.path.relpath(p, os.path.dirname(root)), "library_build_id": t["library_build_id"]}
    return best or {"config": config_id, "value": None, "reason": reason}


def usable_cores():
    """Host cores this process may actually use: the cgroup CPU quota (cpu.max) caps the GPU box's
    share well below os.cpu_count(), and oversubscribing the quota only adds throttling."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


PORTABLE_FLAGS = "-O3 -march=x86-64-v2 -ffp-contract=off -fno-fast-math (prebuilt in the build container, portable across hosts)"


def native_oracle():
    """BASELINE.md times the CPU reference as `-O3 -march=native`; the oracle that travels with the snapshot is built
    -march=x86-64-v2 so that it runs on any host.  For the timed baseline the same source is compiled once more ON the box
    that times it (oracle/_native/, git-ignored scratch; 1-3 s of g++).  Returns (path, flags) or (None, None)."""
    import shutil
    cxx = shutil.which("g++")
    if not cxx:
        return None, None
    odir = os.path.join(ROOT, "oracle")
    out = os.path.join(odir, "_native", "liboracle_native.so")
    flags = ["-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC", "-pthread"]
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run([cxx] + flags + ["-shared", "-o", out, os.path.join(odir, "oracle.cpp")], check=True, timeout=120,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out, " ".join(flags[:4]) + " (compiled on this host)"
    except Exception:  # noqa: BLE001 (no compiler budget, read-only tree, ...: the portable library is the baseline then)
        return None, None


def cpu_baseline(rt, scene, name, nx, ny, max_depth, budget_s=15.0):
    """Times the oracle (kind "port": C++ restatement of the reference, stream RNG order, BVH,
    one worker per host core like threadpool's default) on a bounded sample of the workload:
    the same frame at reduced spp (Mray/s does not depend on spp).  Two builds of the same source are probed at 1 spp —
    the portable library and a -march=native one compiled on this host (BASELINE.md's recipe; on AVX-512 hosts it can be the
    SLOWER of the two: 256-bit code next to libm's legacy-SSE routines) — and the faster one is the baseline."""
    from oracle import binding as orc
    cores = usable_cores()
    opts = orc.options(rng_mode=orc.RNG_STREAM, estimator=orc.EST_RECURSIVE, accel=orc.ACCEL_BVH,
                       n_threads=cores, bvh_seed=1995, bvh_skip_perlin=1)
    probe = rt.make_params(nx, ny, 1, max_depth=max_depth, seed=95)
    native_path, native_flags = native_oracle()
    builds = [(orc.LIB_PATH, PORTABLE_FLAGS)] + ([(native_path, native_flags)] if native_path else [])
    probes = []
    for path, flags in builds:
        try:
            orc.load(path)
            orc.render(scene.flat_ptr, scene.camera, probe, opts)  # (first touch: page the library in)
            t0 = time.time()
            _, _, st = orc.render(scene.flat_ptr, scene.camera, probe, opts)
            probes.append((max(time.time() - t0, 1e-6), path, flags, st.n_rays))
        except Exception:  # noqa: BLE001
            continue
    dt, path, flags, _ = min(probes)
    orc.load(path)
    spp = int(max(1, min(64, budget_s / dt)))
    p = rt.make_params(nx, ny, spp, max_depth=max_depth, seed=95)
    t0 = time.time()
    _, _, st = orc.render(scene.flat_ptr, scene.camera, p, opts)
    dt = max(time.time() - t0, 1e-6)
    orc.load(orc.LIB_PATH)
    return {"value": round(st.n_rays / dt / 1e6, 3), "unit": "Mray/s", "cores": cores, "kind": "port", "flags": flags,
            "builds_probed": {f: round(n / d / 1e6, 2) for d, _, f, n in probes},
            "sample": f"{name} {nx}x{ny}, {spp} spp, max_depth {max_depth}, {st.n_rays} rays in {dt:.1f} s "
                      f"(oracle stream mode: per-column xoshiro256++, recursive estimator, BVH)"}


def in_library_check(rt, scene, renderer, frame, n_dev, timeout_s=150.0):
    """The one-process multi-GPU entry point (rt_multi_create over devices 0..n_dev-1: one host thread per device, RCCL
    ncclAllGather INSIDE the library, de-interleave on device 0) next to the torch.distributed path the bench line times:
    a small frame must equal this process's own rt_render bit for bit (f32 and RGB8, ray counts), then the bench frame is
    rendered once through it for a throughput figure.  Runs in a thread with a deadline: a hang in a collective that has never
    executed with n > 1 must not cost the bench line."""
    import threading

    import numpy as np
    res = {"devices": n_dev, "ok": False, "what": "rt_multi_render: RCCL ncclCommInitAll + ncclAllGather inside librtow_mi355x.so"}

    def work():
        try:
            nx, ny, spp, max_depth = frame
            small = rt.make_params(480, 270, 16, max_depth=max_depth, seed=95)
            ref, ref8, st = renderer.render(scene.camera, small, want_rgb8=True)
            m = rt.MultiRenderer(list(range(n_dev)))
            m.upload(scene)
            img, rgb8, sm = m.render(scene.camera, small, want_rgb8=True)
            res["bit_identical"] = bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgb8, ref8))
            res["rays_equal"] = bool(sm.n_rays == st.n_rays and list(sm.rays_per_depth) == list(st.rays_per_depth))
            full = rt.make_params(nx, ny, spp, max_depth=max_depth, seed=95)
            m.render(scene.camera, full)  # warm-up: buffers
            t0 = time.perf_counter()
            _, _, sf = m.render(scene.camera, full)
            dt = time.perf_counter() - t0
            res["frame"] = f"{nx}x{ny}, {spp} spp to host memory"
            res["value_host_inclusive"] = round(sf.n_rays / dt / 1e6, 3)
            res["ms_per_frame"] = round(dt * 1e3, 3)
            m.close()
            res["ok"] = res["bit_identical"] and res["rays_equal"]
        except Exception as e:  # noqa: BLE001 (reported in the JSON line)
            res["error"] = repr(e)[:400]

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        res["error"], res["hung"] = f"no result after {timeout_s:.0f} s", True
    return res


def first_frame_child(cfg, nx, ny, spp, max_depth, prepare=True):
    """What the reference's own timer covers (utils.rs:15-18 around main.rs:70-126: scene build + render, ONE frame per process),
    measured in a process of its own: HIP runtime start, rt_ctx_create, host-side scene build (image decode included),
    rt_scene_upload and the first rt_render with its buffer allocations, wall clock; a second frame for comparison; and what the
    process allocated on the device.  No torch here: the child is the library and its ctypes wrapper, nothing else.  Prints one
    JSON object."""
    import ctypes
    t0 = time.perf_counter()
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, tot = ctypes.c_size_t(), ctypes.c_size_t()
        return f.value if hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(tot)) == 0 else None
    free0 = free_bytes()
    import ray_tracing_in_one_weekend_amd as rt
    t_rt = time.perf_counter()
    renderer = rt.Renderer(0)
    params = rt.make_params(nx, ny, spp, max_depth=max_depth, seed=95)
    if prepare:  # the frame is known before the world is built (main.rs:64-67): its work buffers are requested meanwhile
        renderer.prepare(params)
    t_ctx = time.perf_counter()
    rt.register_default_images()
    scene = rt.Scene.build(cfg["scene"], nx / ny)
    t_scene = time.perf_counter()
    renderer.upload(scene)
    t_up = time.perf_counter()
    _, _, st1 = renderer.render(scene.camera, params, want_rgb8=True, pinned=True)
    t_f1 = time.perf_counter()
    parts1 = renderer.render_parts()
    free1 = free_bytes()
    _, _, st2 = renderer.render(scene.camera, params, want_rgb8=True, pinned=True)
    t_f2 = time.perf_counter()
    parts2 = renderer.render_parts()
    ms = lambda a, b: round((b - a) * 1e3, 2)
    print(json.dumps({
        "first_frame_ms": ms(t_rt, t_f1),  # rt_ctx_create (+ rt_prepare) + scene build + rt_scene_upload + first rt_render (f32 + RGB8 to the host)
        "rt_prepare": bool(prepare),
        "parts_ms": {"hip_runtime_and_import": ms(t0, t_rt), "rt_ctx_create": ms(t_rt, t_ctx), "scene_build_host": ms(t_ctx, t_scene),
                     "rt_scene_upload": ms(t_scene, t_up), "first_rt_render": ms(t_up, t_f1), "second_rt_render": ms(t_f1, t_f2)},
        # rt_debug_render_parts: where the host spent the first rt_render (allocations one by one, the first kernel launch = code
        # object load + hardware-queue probe, the candidate lists with the one in-frame synchronisation, enqueue, wait)
        "first_render_parts_ms": parts1, "second_render_parts_ms": parts2,
        "first_render_slices": int(st1.n_slices), "second_render_slices": int(st2.n_slices),
        "first_render_device_ms": round(st1.seconds_device * 1e3, 3), "second_render_device_ms": round(st2.seconds_device * 1e3, 3),
        "first_render_trace_launches": int(st1.n_trace_launches), "second_render_trace_launches": int(st2.n_trace_launches),
        "alloc_bytes": (free0 - free1) if free0 is not None and free1 is not None else None,
        "note": "a process of its own (no torch): wall clock from before rt_ctx_create to the first frame in host memory; "
                "alloc_bytes = device memory this process holds after its first frame (hipMemGetInfo before / after)"}))
    renderer.close()


def first_frame(config_id, nx, ny, spp, max_depth, timeout_s=240.0):
    """Runs first_frame_child in a fresh process and returns its record (or {"error": ...})."""
    cmd = [sys.executable, os.path.abspath(__file__), "--first-frame-child", "--config", str(config_id), "--nx", str(nx), "--ny", str(ny),
           "--spp", str(spp), "--max-depth", str(max_depth)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child exited {r.returncode}: {(r.stderr or r.stdout)[-400:]}"}
        return json.loads(lines[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": f"{type(e).__name__}: {e}"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """--gpus N without a torchrun environment: start the N rank processes as a child (this process has not touched a
    GPU and never will; it only relays the child's output and exit code)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def init_ranks(dist, backend, device, rank, world, emit, timeout_s=None):
    """init_process_group with a limit of its own.  The rendezvous honours `timeout`; a communicator bring-up that wedges inside
    the library (RCCL's eager init with device_id) does not, so a watchdog thread ends the rank: a JSON line with "error" from
    rank 0 (stderr from the others) and a non-zero exit, instead of hanging until the driver's limit kills the run."""
    import datetime
    import threading
    timeout_s = float(os.environ.get("RTOW_INIT_TIMEOUT_S", timeout_s or 180.0))
    done = threading.Event()

    def give_up(reason):
        msg = (f"bench.py: rank {rank} of {world}: init_process_group({backend}) {reason} "
               f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')}, MASTER_PORT={os.environ.get('MASTER_PORT')}, "
               f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})")
        sys.stderr.write(msg + "\n")
        sys.stderr.flush()
        if rank == 0:
            emit({"error": msg, "value": None, "n_gpus": world, "rendered": False})
        os._exit(3)

    def watchdog():
        if not done.wait(timeout_s + 15.0):
            give_up(f"did not return within {timeout_s + 15.0:.0f} s")

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        kw = {"device_id": device} if device is not None else {}
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    except Exception as e:  # noqa: BLE001 — whatever the rendezvous raised is the diagnostic
        done.set()
        give_up(f"failed: {type(e).__name__}: {e}")
    done.set()


def trace_roofline(trace_bytes, trace_s):
    achieved = trace_bytes / max(trace_s, 1e-12) / 1e9
    return round(achieved, 2), round(achieved / HBM_PEAK_GBPS, 5)


def build_record(*, config_id, cfg, nx, ny, spp, spp_total, scaling, max_depth, band, world, backend, steps, warmup, elapsed_max,
                 rays_total, gather_ms, per_rank, rank0, build_id, rendered=True, profiles_root=None):
    """The JSON line of rank 0 from the aggregated measurements (pure: the CPU test of the multi-rank launch path builds the
    same line from a rehearsal's numbers).  per_rank: [(trace_seconds, trace_bytes_algorithmic)] of every rank over the timed
    steps; rank0: dict of rank 0's last RtStats fields + launch totals (None when nothing was rendered)."""
    gather_name = "RCCL gather to rank 0 over xGMI" if backend == "nccl" else f"{backend} gather to rank 0 through host memory (rehearsal)"
    per_gpu = f"{spp} spp per GPU ({spp_total} spp total)" if scaling == "weak" else f"{spp_total} spp"
    workload = (f"config {config_id}: {cfg['what']} {nx}x{ny}, {per_gpu}, max_depth {max_depth}, "
                f"seed 95, counter RNG; rows sharded in bands of {band} over {world} GPU(s)"
                + (f", {gather_name} of the f32 framebuffer per step" if world > 1 else ""))
    r0 = rank0 or {}
    trace_s, trace_bytes = per_rank[0] if per_rank else (0.0, 0)
    achieved, frac = trace_roofline(trace_bytes, trace_s) if rendered else (None, None)
    launches = r0.get("launches", 0)
    traffic, traffic_src = pmc_traffic(workload, build_id, trace_bytes / max(launches, 1)) if rendered else (None, None)
    out = {
        # BASELINE.json's metric is quoted on config 2; the other configs carry their own frame in the label
        "metric": f"Mray/s (primary+secondary) at {nx}x{ny}/{spp_total}spp",
        "value": round(rays_total / max(elapsed_max, 1e-12) / 1e6, 3) if rendered else None,
        "unit": "Mray/s",
        "n_gpus": world,
        "rccl_ranks": world if (backend == "nccl" and world > 1) else 0,  # ranks of the RCCL communicator the gather ran on
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed_max / max(steps, 1) * 1e3, 3) if rendered else None,
        "gather_ms": gather_ms,  # gather to rank 0 + its de-interleave per step, HIP events on the launch stream, max over ranks (null on 1 GPU)
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload,
                   "paths_per_step": nx * ny * spp_total, "rays_per_step_rank0": int(r0.get("n_rays", 0)),
                   "rays_per_path": round(r0.get("n_rays", 0) / max(r0.get("n_paths", 0), 1), 4),
                   "texture_fetches_per_step_rank0": int(r0.get("n_texture_fetches", 0)),
                   "spp_slices": int(r0.get("n_slices", 0))},
        "roofline": {"kernel": "trace step = closest hit (k_intersect / k_intersect_grid) + k_shade (the survey's k_trace_shade, split)",
                     "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac,
                     # every rank's own trace step (rank order): a curve over N carries its HBM fraction per GPU
                     "per_rank": [dict(zip(("achieved", "frac"), trace_roofline(b, t))) for t, b in per_rank] if rendered else [],
                     "traffic": traffic, "traffic_source": traffic_src,
                     "valu": pmc_valu(workload, build_id) if rendered else None,
                     "bytes_per_launch": round(trace_bytes / max(launches, 1), 1),
                     "avg_launch_us": round(trace_s / max(launches, 1) * 1e6, 3),
                     "launches": launches,
                     "note": "per launch = one kernel of one depth over all shards (2 per depth; each runs as two concurrent half-grid dispatches on two "
                             "streams, so rocprof lists twice as many dispatches of about this duration); algorithmic bytes = "
                             "48 B/ray read + 48 B/surviving ray written + 12 B/path radiance + 12 B/ImageTex fetch (SURVEY.md 8(d): "
                             "96 B/ray + 24 B/path + 12 B/fetch over gen+trace+resolve); time = HIP events around the trace launches "
                             "of every slice on the launch stream; traffic = PMC FETCH_SIZE*2 + WRITE_SIZE of both kernels per launch "
                             "(x2 calibrated for streams AND gathers: profiles/round3/fetch_calibration.json).  The algorithmic figure "
                             "follows the SURVEY 8(d) CONVENTION of a 48 B ray record; the records this layout moves are 40 B + an 8 B "
                             "hit record between the kernels, i.e. the same 48 B per ray read and 40 B per survivor written"},
        "library_build_id": build_id,
    }
    if world > 1:
        # the same frame on ONE GPU with these kernels (committed record): efficiency = value / (n_gpus x this value)
        out["single_gpu_reference"] = single_gpu_reference(config_id, out["metric"], build_id, profiles_root)
    if rendered:
        out["whole_path"] = {"bytes_algorithmic_per_step": int(r0["bytes_algorithmic"]),
                             "device_seconds_per_step": round(r0["seconds_device"], 6),
                             "hbm_frac": round(r0["bytes_algorithmic"] / max(r0["seconds_device"], 1e-12) / 1e9 / HBM_PEAK_GBPS, 5)}
    else:
        out["rendered"] = False
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0] + sorted(CONFIGS),
                    help="BASELINE.json config; default: 2 on one GPU, 3 (the 8-GPU frame, strong scaling) with --gpus N > 1")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None)
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel (weak scaling: per GPU)")
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--spp-slice", type=int, default=0)
    ap.add_argument("--band", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="only the warm-up and the timed steps: no CPU baseline, no host-inclusive leg, no in-library check, no second "
                         "workload (the runs that rocprofv3 counts: scripts/collect_traffic.py, collect_valu.py, the kernel statistics)")
    ap.add_argument("--in-library", action="store_true",
                    help="after the timed region, rank 0 also renders through rt_multi_render (one process, all --gpus devices, the RCCL "
                         "gather inside the library) and reports bit-identity with rt_render; always on when --gpus > 1")
    ap.add_argument("--first-frame-child", action="store_true", help=argparse.SUPPRESS)  # (the fresh process of first_frame())
    ap.add_argument("--no-prepare", action="store_true", help=argparse.SUPPRESS)  # (first-frame child without the rt_prepare hint)
    ap.add_argument("--launcher-check", action="store_true",
                    help="rendezvous, world-size assertion and one framebuffer gather of a synthetic band buffer; no rendering "
                         "(the CPU test of the multi-rank launch path: RTOW_DIST_BACKEND=gloo, no GPU needed)")
    args = ap.parse_args()

    if args.first_frame_child:
        cfg = CONFIGS[args.config or 2]
        first_frame_child(cfg, args.nx or cfg["nx"], args.ny or cfg["ny"], args.spp or cfg["spp"], args.max_depth, prepare=not args.no_prepare)
        return
    n_req = max(args.gpus, 1)
    if n_req > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(n_req))

    # This process's stdout carries exactly ONE line, the JSON record of rank 0.  Libraries print banners there (RCCL its version
    # when a communicator comes up, gloo its peer count): file descriptor 1 points at stderr from here on, and the record goes to
    # the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(record) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The reference renders ONE frame per process (main.rs:62-129) and its timer covers scene build + render (utils.rs:15-18):
    # that frame is measured in a child process, twice — here, BEFORE this process has imported torch or touched the GPU (a
    # quiet device: nobody else holds or has just freed memory), and after the timed region while this process still holds its
    # own work buffers (a device with a tenant).  The second is `first_frame`, the first `first_frame_quiet_device`.
    ff_quiet = None
    if world == 1 and not (args.timed_only or args.launcher_check):
        c0 = CONFIGS[args.config or 2]
        ff_quiet = first_frame(args.config or 2, args.nx or c0["nx"], args.ny or c0["ny"], args.spp or c0["spp"], args.max_depth)

    import torch
    import torch.distributed as dist

    if world != n_req:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torchrun --nproc-per-node {n_req}, or let `python bench.py --gpus {n_req}` spawn them)")
    # RTOW_DIST_BACKEND=gloo: rehearsal of the multi-process path on a box with fewer GPUs than ranks
    # (ranks share devices, the gather goes through host memory); the real run is nccl == RCCL over xGMI
    backend = os.environ.get("RTOW_DIST_BACKEND", "nccl")
    # the workload: config 2 on one GPU (BASELINE.json's metric), config 3 — the frame north_star names for the 8 GPUs of a node,
    # split N ways — when there are several and nothing else was asked for
    config_id = args.config or (2 if world == 1 else 3)
    cfg = CONFIGS[config_id]
    nx, ny = args.nx or cfg["nx"], args.ny or cfg["ny"]
    spp = args.spp or cfg["spp"]
    scaling = args.scaling or cfg["scaling"]
    spp_total = spp * world if scaling == "weak" else spp
    common = dict(config_id=config_id, cfg=cfg, nx=nx, ny=ny, spp=spp, spp_total=spp_total, scaling=scaling, max_depth=args.max_depth,
                  band=args.band, world=world, steps=args.steps, warmup=args.warmup)

    if args.launcher_check:
        from ray_tracing_in_one_weekend_amd import shard
        # the gather runs over RCCL only when the backend is nccl AND every rank has a device of its own; otherwise gloo on
        # host tensors (NCCL has no CPU backend), and the line says so
        on_rccl = backend == "nccl" and torch.cuda.is_available() and torch.cuda.device_count() >= world
        dev = torch.device("cpu")
        if on_rccl:
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
        if world > 1:
            init_ranks(dist, "nccl" if on_rccl else "gloo", dev if on_rccl else None, rank, world, emit)
            assert dist.get_world_size() == n_req
        cny, cnx = 64, 16
        rows = shard.shard_rows(cny, args.band, world, rank)
        local = torch.zeros((len(rows), cnx, 3), dtype=torch.float32, device=dev)
        local[:, :, 0] = torch.as_tensor(rows, dtype=torch.float32, device=dev)[:, None]  # every pixel carries its image row
        t0 = time.perf_counter()
        full = shard.gather_framebuffer(local, cny, args.band, dst=0) if world > 1 else local
        gather_ms = round((time.perf_counter() - t0) * 1e3, 3) if world > 1 else None
        # only rank 0 holds the frame (the others sent their bands and got None back)
        ok = (full is None) if rank else bool((full[:, 0, 0].cpu() == torch.arange(cny, dtype=torch.float32)).all())
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        if world > 1:
            dist.barrier()
        if rank == 0:
            ran = "nccl" if (on_rccl and world > 1) else "gloo"
            # the line a rendering run of these arguments would print, with nothing rendered: same keys, the launch path's own numbers
            rec = build_record(**common, backend=ran, elapsed_max=0.0, rays_total=0.0, gather_ms=gather_ms,
                               per_rank=[(0.0, 0)] * world, rank0=None, build_id=None, rendered=False)
            rec.update({"launcher_check": ok, "rccl_ranks": world if ran == "nccl" else 0, "backend": ran,
                        "gather": "RCCL gather to rank 0 over xGMI" if ran == "nccl" else "gloo gather to rank 0 through host memory (rehearsal)"})
            emit(rec)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    device_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        init_ranks(dist, backend, torch.device("cuda", device_index) if backend == "nccl" else None, rank, world, emit)
        assert dist.get_world_size() == n_req, (dist.get_world_size(), n_req)

    import ray_tracing_in_one_weekend_amd as rt
    from ray_tracing_in_one_weekend_amd import shard

    rt.register_default_images()
    renderer = rt.Renderer(device_index)  # raises if librtow_mi355x.so is missing
    # the frame is known before the world is built (main.rs:64-67): the context requests its work buffers meanwhile
    renderer.prepare(rt.make_params(nx, ny, spp_total, max_depth=args.max_depth, seed=95, shard_band=args.band, shard_count=world,
                                    shard_id=rank, spp_slice=args.spp_slice))
    # a dedicated non-default stream: the library's launches, its HIP events and the RCCL gather are
    # all ordered on it (stream handle 0 would make the library fall back to its own stream); the band
    # buffer is allocated on that stream too
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        dist.all_reduce(c, op=op)
        return c

    def run_workload(scene, w_nx, w_ny, w_spp_total, steps, warmup):
        """warm-up, then exactly `steps` timed steps between fences; returns the aggregated measurements of all ranks"""
        params = rt.make_params(w_nx, w_ny, w_spp_total, max_depth=args.max_depth, seed=95, shard_band=args.band,
                                shard_count=world, shard_id=rank, spp_slice=args.spp_slice)
        rows = renderer.shard_rows(params)
        local = torch.zeros((rows, w_nx, 3), dtype=torch.float32, device="cuda")
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if world > 1 else []

        def step(want_stats, k=None):
            st = renderer.render_device(scene.camera, params, local.data_ptr(), stream=stream, want_stats=want_stats)
            if world > 1 and backend == "nccl":
                if k is not None:
                    ev[k][0].record()
                full = shard.gather_framebuffer(local, w_ny, args.band, dst=0)  # RCCL gather; rank 0 de-interleaves
                if k is not None:
                    ev[k][1].record()
            elif world > 1:
                torch.cuda.current_stream().synchronize()
                t0 = time.perf_counter()
                full = shard.gather_framebuffer(local.cpu(), w_ny, args.band, dst=0)
                if k is not None:
                    ev[k] = (time.perf_counter() - t0) * 1e3
            else:
                full = local
            return st, full

        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        stats = []
        for k in range(steps):
            st, _ = step(True, k)  # reading the counters synchronises; the counters are part of the metric
            stats.append(st)
        fence()
        elapsed = time.perf_counter() - t0
        gather_ms_local = 0.0
        if world > 1:
            gather_ms_local = sum((e if isinstance(e, float) else e[0].elapsed_time(e[1])) for e in ev) / max(steps, 1)
        trace_s = sum(s.seconds_trace for s in stats)
        trace_bytes = float(sum(s.bytes_trace_algorithmic for s in stats))
        t = torch.tensor([elapsed, float(sum(s.n_rays for s in stats)), gather_ms_local], dtype=torch.float64, device="cuda")
        per_rank = [(trace_s, trace_bytes)]
        if world > 1:
            tmax = reduce_(t.clone(), dist.ReduceOp.MAX)
            tsum = reduce_(t.clone(), dist.ReduceOp.SUM)
            elapsed_max, rays_total, gather_ms = tmax[0].item(), tsum[1].item(), round(tmax[2].item(), 3)
            mine = torch.tensor([trace_s, trace_bytes], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            per_rank = [(float(q[0]), float(q[1])) for q in parts]
        else:
            elapsed_max, rays_total, gather_ms = elapsed, t[1].item(), None
        s0 = stats[-1]
        rank0 = {"n_paths": s0.n_paths, "n_rays": s0.n_rays, "n_texture_fetches": s0.n_texture_fetches, "n_slices": s0.n_slices,
                 "bytes_algorithmic": s0.bytes_algorithmic, "seconds_device": s0.seconds_device,
                 "launches": sum(s.n_trace_launches for s in stats)}
        del local
        return dict(elapsed_max=elapsed_max, rays_total=rays_total, gather_ms=gather_ms, per_rank=per_rank, rank0=rank0), params

    scene = rt.Scene.build(cfg["scene"], nx / ny)
    renderer.upload(scene)
    meas, params = run_workload(scene, nx, ny, spp_total, args.steps, args.warmup)

    # With several GPUs and the default workload, the one-GPU headline frame rides along: config 2, weak scaling (every rank
    # 1080 / N rows x 256 N samples), two timed steps — the curve the round-3 lines were on
    also = None
    if world > 1 and not args.config and not args.timed_only:
        c2 = CONFIGS[2]
        scene2 = rt.Scene.build(c2["scene"], c2["nx"] / c2["ny"])
        renderer.upload(scene2)
        m2, _ = run_workload(scene2, c2["nx"], c2["ny"], c2["spp"] * world, 2, 1)
        also = (c2, m2)
        renderer.upload(scene)

    if world > 1:  # every collective of the measurement is behind us: the other ranks leave and free their GPUs
        dist.barrier()
        dist.destroy_process_group()
    in_library = None
    if rank == 0 and (args.in_library or world > 1) and not args.timed_only:
        in_library = in_library_check(rt, scene, renderer, (nx, ny, spp_total, args.max_depth), world)
    if rank == 0:
        build_id = renderer.build_id
        out = build_record(**common, backend=backend, build_id=build_id, **meas)
        if also:
            c2, m2 = also
            r2 = build_record(config_id=2, cfg=c2, nx=c2["nx"], ny=c2["ny"], spp=c2["spp"], spp_total=c2["spp"] * world, scaling="weak",
                              max_depth=args.max_depth, band=args.band, world=world, backend=backend, steps=2, warmup=1, build_id=build_id, **m2)
            out["also"] = {k: r2[k] for k in ("metric", "value", "unit", "ms_per_step", "gather_ms", "scaling", "steps", "warmup")}
            out["also"]["workload"] = r2["config"]["workload"]
            out["also"]["roofline"] = {k: r2["roofline"][k] for k in ("achieved", "frac", "per_rank")}
            # weak scaling: every rank does the work of the one-GPU headline frame, whose committed line is the reference
            out["also"]["single_gpu_reference"] = single_gpu_reference(2, f"Mray/s (primary+secondary) at {c2['nx']}x{c2['ny']}/{c2['spp']}spp", build_id)
        if world == 1 and not args.timed_only:
            # the same frame handed to the HOST as the reference's output is (f32 frame + flipped RGB8 through rt_render into
            # page-locked memory of rt_host_alloc: the D2H copies included), never `value`: reported beside it
            renderer.render(scene.camera, params, want_rgb8=True, pinned=True)  # (the staging buffers' first touch)
            th0 = time.perf_counter()
            host_rays = 0
            for _ in range(max(1, min(args.steps, 3))):
                _, _, sth = renderer.render(scene.camera, params, want_rgb8=True, pinned=True)
                host_rays += sth.n_rays
            out["value_host_inclusive"] = round(host_rays / (time.perf_counter() - th0) / 1e6, 3)
        if world == 1 and not args.timed_only:
            # every number above is a warm, steady-state frame; the reference renders ONE frame per process (main.rs:62-129)
            ff = first_frame(config_id, nx, ny, spp_total, args.max_depth)
            out["first_frame"] = ff
            out["first_frame_quiet_device"] = ff_quiet
            out["first_frame_ms"] = ff.get("first_frame_ms")
            out["alloc_bytes"] = ff.get("alloc_bytes")
        if args.in_library or world > 1:
            out["in_library"] = in_library
        if world == 1 and not args.no_cpu_baseline and not args.timed_only:
            out["cpu_baseline"] = cpu_baseline(rt, scene, cfg["scene"], nx, ny, args.max_depth)
            out["speedup_vs_cpu_baseline"] = round(out["value"] / max(out["cpu_baseline"]["value"], 1e-9), 2)
        emit(out)
        if in_library and in_library.get("hung"):
            os._exit(0)  # the line is out; do not wait for a wedged collective at interpreter exit
    renderer.close()


if __name__ == "__main__":
    main()
