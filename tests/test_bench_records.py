"""bench.py's use of committed measurement records (no GPU): a PMC profile is quoted only for the workload AND the library build
it was taken on."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_profile_is_refused_across_builds():
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "traffic_config2.json")))
    newest = json.load(open(files[-1]))
    workload, build = newest["bench_config"]["workload"], newest["library_build_id"]
    assert build and len(build) == 16
    traffic, src = bench.pmc_traffic(workload, build)
    assert traffic == newest["trace_step_bytes_per_launch"] and src.endswith("traffic_config2.json")
    traffic, why = bench.pmc_traffic(workload, "0123456789abcdef")  # other kernels: no number, and the reason says so
    assert traffic is None and "was taken on build" in why
    assert bench.pmc_traffic("some other workload", build) == (None, None)
    assert bench.pmc_valu(workload, "0123456789abcdef") is None
    v = bench.pmc_valu(workload, build)
    assert v and 0.0 < v["k_intersect"]["issue_frac"] <= 1.0 and 0.0 < v["k_shade"]["lane_util"] <= 1.0


def test_build_id_covers_sources_and_flags():
    from ray_tracing_in_one_weekend_amd import build as b
    a = b.gpu_build_id()
    assert len(a) == 16 and a == b.gpu_build_id()
    saved = list(b.HIPCC_FLAGS)
    try:
        b.HIPCC_FLAGS.append("-DRT_SOMETHING")
        assert b.gpu_build_id() != a
    finally:
        b.HIPCC_FLAGS[:] = saved
