"""The C-ABI libraries load on a machine without a GPU and export every symbol the headers in
include/ declare; struct layouts seen by ctypes match the C compiler's."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef RT_PROFILE_LANES.*?#endif", "", src, flags=re.S)  # diagnostic builds only, absent from the product library
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?\s+\*?\s*((?:rt|rth)_[a-z0-9_]+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_gpu_library_exports_every_declared_symbol(rt):
    lib = rt._ffi.load_gpu_library()
    names = _declared_functions("rtow_mi355x.h")
    assert set(names) == set(rt._ffi.GPU_SYMBOLS), (names, rt._ffi.GPU_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    # ... and nothing else with C linkage: exported == declared, debug entry points included
    nm = subprocess.run(["nm", "-D", "--defined-only", rt._ffi.GPU_LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted(ln.split()[-1] for ln in nm.splitlines() if re.search(r" T rt_[a-z0-9_]+$", ln))
    assert exported == names, set(exported) ^ set(names)
    assert lib.rt_abi_version() == 9 == rt._ffi.EXPECTED_ABI
    assert len(lib.rt_build_id()) in (7, 16)  # "unknown" or 16 hex digits


def test_host_library_exports_every_declared_symbol(rt):
    lib = rt._ffi.load_host_library()
    names = _declared_functions("rtow_host.h")
    assert set(names) == set(rt._ffi.HOST_SYMBOLS), (names, rt._ffi.HOST_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layouts_match_the_c_compiler(rt, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rtow_mi355x.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(RtFlatScene),sizeof(RtCamera),sizeof(RtParams),sizeof(RtStats),sizeof(RtBounceIO),'
                   'offsetof(RtFlatScene,sky_type),offsetof(RtStats,rays_per_depth),offsetof(RtParams,seed));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    f = rt._ffi
    want = [C.sizeof(f.RtFlatScene), C.sizeof(f.RtCamera), C.sizeof(f.RtParams), C.sizeof(f.RtStats), C.sizeof(f.RtBounceIO),
            f.RtFlatScene.sky_type.offset, f.RtStats.rays_per_depth.offset, f.RtParams.seed.offset]
    assert got == want


def test_shard_helpers_need_no_gpu(rt):
    lib = rt._ffi.load_gpu_library()
    from ray_tracing_in_one_weekend_amd import shard
    for ny, band, world in ((1080, 8, 8), (225, 8, 3), (7, 4, 2), (100, 1, 4)):
        total = 0
        for r in range(world):
            rows = shard.shard_rows(ny, band, world, r)
            assert lib.rt_shard_rows(ny, band, world, r) == len(rows)
            assert [lib.rt_shard_row_to_image_row(k, band, world, r) for k in range(len(rows))] == rows.tolist()
            total += len(rows)
        assert total == ny
    assert lib.rt_shard_rows(10, 0, 1, 0) == 10


def test_context_creation_fails_loudly_without_a_gpu(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError, match="rt_ctx_create"):
        rt.Renderer(0)


def test_the_libraries_read_no_environment_variable():
    """Configuration goes through the C-ABI (rt_debug_set_option: per context), never through the process environment."""
    for sub in ("csrc", "host"):
        d = os.path.join(ROOT, "ray_tracing_in_one_weekend_amd", sub)
        for fn in os.listdir(d):
            assert "getenv" not in open(os.path.join(d, fn), errors="ignore").read(), fn


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ray_tracing_in_one_weekend_amd")
    for dp, _, fs in os.walk(pkg):
        for fn in fs:
            if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn
                assert not re.search(r'#include\s+"[^"]*oracle', txt), fn
