"""N > 1 path on CPU: two gloo ranks each produce their row-interleaved shard (the CPU oracle
stands in for the GPU renderer here — it is only the checker's pixel source), the framebuffer
gather of ray_tracing_in_one_weekend_amd.shard reassembles the frame, and the result must equal
the unsharded render bit for bit on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, band, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import ray_tracing_in_one_weekend_amd as rt
    from oracle import binding as orc
    from ray_tracing_in_one_weekend_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt.register_default_images()
    scene = rt.Scene.build("sphere_scene", 96 / 54)
    nx, ny = 96, 54
    p = rt.make_params(nx, ny, 4, max_depth=12, shard_band=band, shard_count=world, shard_id=rank)
    local, _, st = orc.render(scene.flat_ptr, scene.camera, p, orc.options(n_threads=2))
    full = shard.gather_framebuffer(torch.from_numpy(local), ny, band)
    # dst=0: only rank 0 receives and de-interleaves (what bench.py does every step); the other ranks get None
    only0 = shard.gather_framebuffer(torch.from_numpy(local), ny, band, dst=0)
    assert (only0 is None) == (rank != 0)
    if rank == 0:
        assert torch.equal(only0, full)
    rays = torch.tensor([st.n_rays], dtype=torch.int64)
    dist.all_reduce(rays)
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"rays_{rank}.npy"), rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("band", [8, 5])
def test_two_rank_gloo_gather_reassembles_the_frame(rt, orc, tmp_path, band):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, band, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    scene = rt.Scene.build("sphere_scene", 96 / 54)
    ref, _, st = orc.render(scene.flat_ptr, scene.camera, rt.make_params(96, 54, 4, max_depth=12), orc.options(n_threads=2))
    for r in range(world):
        full = np.load(tmp_path / f"full_{r}.npy")
        assert full.shape == ref.shape and np.array_equal(full, ref)
        assert int(np.load(tmp_path / f"rays_{r}.npy")[0]) == st.n_rays


def test_deinterleave_roundtrip_uneven_rows():
    from ray_tracing_in_one_weekend_amd import shard
    ny, nx, band, world = 37, 5, 4, 3
    img = np.arange(ny * nx * 3, dtype=np.float32).reshape(ny, nx, 3)
    pad = shard.max_shard_rows(ny, band, world)
    parts = []
    for r in range(world):
        rows = shard.shard_rows(ny, band, world, r)
        buf = np.zeros((pad, nx, 3), np.float32)
        buf[:len(rows)] = img[rows]
        parts.append(buf)
    assert np.array_equal(shard.deinterleave(parts, ny, band, world), img)


def test_bench_launcher_spawns_the_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment must start two rank processes itself (before it touches a
    GPU), rendezvous on 127.0.0.1, see world size 2 and gather a band buffer.  --launcher-check does exactly that without
    rendering (no GPU here); gloo stands in for RCCL."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RTOW_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-check", "--band", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["launcher_check"] is True and out["n_gpus"] == 2 and out["backend"] == "gloo"
    # the line has the shape of the one a rendering 2-GPU run prints: with no --config it is BASELINE's config 3 — the frame
    # north_star names for the GPUs of a node — under strong scaling, with the gather time and the HBM fraction per rank
    assert out["config"]["workload"].startswith("config 3:") and "3840x2160" in out["config"]["workload"] and out["scaling"] == "strong"
    assert out["metric"].endswith("3840x2160/1024spp") and out["rendered"] is False and out["value"] is None
    assert "frac" in out["roofline"] and "per_rank" in out["roofline"] and out["gather_ms"] > 0
    # a rank count that does not match --gpus is refused, not silently benchmarked as something else
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-check"], env=env2,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_rank_that_cannot_rendezvous_reports_and_exits(tmp_path):
    """A rank whose peers never arrive must not hang to the driver's limit: init_ranks gives up after its own timeout, rank 0
    prints a JSON line with "error" and the process exits non-zero."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RTOW_DIST_BACKEND="gloo", RTOW_INIT_TIMEOUT_S="5", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-check"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stderr[-2000:]
    out = json.loads(lines[-1])
    assert "init_process_group(gloo)" in out["error"] and out["value"] is None and out["n_gpus"] == 2


def test_bench_record_of_a_two_rank_run():
    """bench.build_record on the numbers of a 2-rank run: whole-job rays over the slowest rank's time, every rank's own
    trace-step HBM fraction, the gather time; and a one-GPU default run is config 2."""
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS[3]
    r0 = {"n_paths": 10, "n_rays": 25, "n_texture_fetches": 0, "n_slices": 1, "bytes_algorithmic": 4000, "seconds_device": 1e-6, "launches": 4}
    rec = bench.build_record(config_id=3, cfg=cfg, nx=3840, ny=2160, spp=1024, spp_total=1024, scaling="strong", max_depth=50, band=8, world=2,
                             backend="nccl", steps=2, warmup=1, elapsed_max=0.5, rays_total=4e9, gather_ms=0.4,
                             per_rank=[(0.2, 4e11), (0.25, 4e11)], rank0=r0, build_id="0" * 16)
    assert rec["value"] == 8000.0 and rec["ms_per_step"] == 250.0 and rec["gather_ms"] == 0.4 and rec["rccl_ranks"] == 2
    assert rec["roofline"]["frac"] == 0.25 and [q["frac"] for q in rec["roofline"]["per_rank"]] == [0.25, 0.2]
    assert "RCCL gather to rank 0" in rec["config"]["workload"] and rec["config"]["paths_per_step"] == 3840 * 2160 * 1024


def test_bench_record_carries_the_one_gpu_figure_of_the_same_frame(tmp_path):
    """An N > 1 line names the committed one-GPU record of the same frame on the same kernels ("single_gpu_reference"), so that
    scaling efficiency = value / (N x reference) is computable from the line alone; a record of another build, of another frame or
    of several GPUs is refused with the reason, and a one-GPU line has no such key."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    cfg = bench.CONFIGS[3]
    metric = "Mray/s (primary+secondary) at 3840x2160/1024spp"
    rd = tmp_path / "profiles" / "round9"
    rd.mkdir(parents=True)
    common = dict(config_id=3, cfg=cfg, nx=3840, ny=2160, spp=1024, spp_total=1024, scaling="strong", max_depth=50, band=8, backend="nccl", steps=2,
                  warmup=1, elapsed_max=0.5, rays_total=4e9, gather_ms=0.4, rank0={"n_paths": 10, "n_rays": 25, "n_texture_fetches": 0, "n_slices": 1,
                                                                                   "bytes_algorithmic": 4000, "seconds_device": 1e-6, "launches": 4},
                  profiles_root=str(tmp_path / "profiles"))
    rec = bench.build_record(**common, world=2, per_rank=[(0.2, 4e11), (0.25, 4e11)], build_id="a" * 16)
    assert rec["single_gpu_reference"]["value"] is None and "no profiles" in rec["single_gpu_reference"]["reason"]
    (rd / "bench_config3.json").write_text(json.dumps({"metric": metric, "value": 5000.0, "unit": "Mray/s", "n_gpus": 1, "ms_per_step": 800.0,
                                                       "library_build_id": "a" * 16}))
    rec = bench.build_record(**common, world=2, per_rank=[(0.2, 4e11), (0.25, 4e11)], build_id="a" * 16)
    ref = rec["single_gpu_reference"]
    assert ref["value"] == 5000.0 and ref["source"] == "profiles/round9/bench_config3.json" and ref["library_build_id"] == "a" * 16
    assert rec["value"] / (rec["n_gpus"] * ref["value"]) == 0.8  # the efficiency a reader computes
    rec = bench.build_record(**common, world=2, per_rank=[(0.2, 4e11), (0.25, 4e11)], build_id="b" * 16)  # other kernels
    assert rec["single_gpu_reference"]["value"] is None and "was taken on build " + "a" * 16 in rec["single_gpu_reference"]["reason"]
    (rd / "bench_config3.json").write_text(json.dumps({"metric": metric.replace("1024spp", "256spp"), "value": 5000.0, "n_gpus": 1, "library_build_id": "a" * 16}))
    rec = bench.build_record(**common, world=2, per_rank=[(0.2, 4e11), (0.25, 4e11)], build_id="a" * 16)
    assert rec["single_gpu_reference"]["value"] is None and "not a one-GPU line of this frame" in rec["single_gpu_reference"]["reason"]
    rec = bench.build_record(**common, world=1, per_rank=[(0.2, 4e11)], build_id="a" * 16)
    assert "single_gpu_reference" not in rec


def test_bench_first_frame_leg_reports_instead_of_raising():
    """bench.first_frame runs the cold frame (context + scene build + upload + first render) in a process of its own; where that
    process cannot render (no GPU here) the bench line gets an "error" entry under "first_frame", never an exception or a hang."""
    sys.path.insert(0, ROOT)
    import bench
    ff = bench.first_frame(2, 64, 36, 1, 4, timeout_s=120.0)
    assert isinstance(ff, dict) and ("error" in ff or ff.get("first_frame_ms", 0) > 0)
