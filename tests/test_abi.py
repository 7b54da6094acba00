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


PRODUCT_HEADER, DEBUG_HEADER = "rtow_mi355x.h", "rtow_mi355x_debug.h"


def test_gpu_library_exports_every_declared_symbol(rt):
    lib = rt._ffi.load_gpu_library()
    product, hooks = _declared_functions(PRODUCT_HEADER), _declared_functions(DEBUG_HEADER)
    # the header a host binds holds lifecycle / scene / render / multi-GPU / progress and no test hook
    assert product and not [n for n in product if "debug" in n or n == "rt_get_depth_timings"], product
    assert hooks and not set(product) & set(hooks)
    names = sorted(product + hooks)
    assert set(names) == set(rt._ffi.GPU_SYMBOLS), (names, rt._ffi.GPU_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    # ... and nothing else with C linkage: exported == declared, debug entry points included
    nm = subprocess.run(["nm", "-D", "--defined-only", rt._ffi.GPU_LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted(ln.split()[-1] for ln in nm.splitlines() if re.search(r" T rt_[a-z0-9_]+$", ln))
    assert exported == names, set(exported) ^ set(names)
    assert lib.rt_abi_version() == 11 == rt._ffi.EXPECTED_ABI
    assert len(lib.rt_build_id()) in (7, 16)  # "unknown" or 16 hex digits


def test_host_library_exports_every_declared_symbol(rt):
    lib = rt._ffi.load_host_library()
    names = _declared_functions("rtow_host.h")
    assert set(names) == set(rt._ffi.HOST_SYMBOLS), (names, rt._ffi.HOST_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_struct_layouts_match_the_c_compiler(rt, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rtow_mi355x_debug.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(RtFlatScene),sizeof(RtCamera),sizeof(RtParams),sizeof(RtStats),sizeof(RtBounceIO),'
                   'offsetof(RtFlatScene,sky_type),offsetof(RtStats,rays_per_depth),offsetof(RtParams,seed));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    f = rt._ffi
    want = [C.sizeof(f.RtFlatScene), C.sizeof(f.RtCamera), C.sizeof(f.RtParams), C.sizeof(f.RtStats), C.sizeof(f.RtBounceIO),
            f.RtFlatScene.sky_type.offset, f.RtStats.rays_per_depth.offset, f.RtParams.seed.offset]
    assert got == want


# ---- the Rust binding text of INTEGRATION.md section 1 against the headers -----------------------------------------------
_C2RUST = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int": "i32", "float": "f32", "double": "f64",
           "size_t": "usize", "char": "c_char", "void": "c_void"}


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef RT_PROFILE_LANES.*?#endif", "", src, flags=re.S)
    return src


def _c_type_to_rust(ctype, array=None):
    """'const float*' -> '*const f32', 'RtCtx**' -> '*mut *mut RtCtx', 'float' + [3] -> '[f32; 3]' (a parameter array decays)."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct")]
    assert len(base) == 1, ctype
    r = _C2RUST.get(base[0], base[0])
    for k in range(stars):
        r = ("*const " if (const and k == 0) else "*mut ") + r
    return f"[{r}; {array}]" if array else r


def _c_structs(header):
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", _strip_c(open(os.path.join(ROOT, "include", header)).read()), flags=re.S):
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            m = re.match(r"(.*?[\s\*])(\w+(?:\s*,\s*\w+)*)\s*(?:\[(\d+)\])?$", decl, flags=re.S)
            assert m, decl
            for fname in [x.strip() for x in m.group(2).split(",")]:
                fields.append((fname, _c_type_to_rust(m.group(1), m.group(3))))
        out[name] = fields
    return out


def _c_prototypes(header):
    out = {}
    src = _strip_c(open(os.path.join(ROOT, "include", header)).read())
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w+\s*\*?)\s*((?:rt)_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        params = []
        if args.strip() != "void":
            for a in [x.strip() for x in args.split(",")]:
                if a.startswith("RtProgressFn"):
                    params.append(("fn", "RtProgressFn"))
                    continue
                m = re.match(r"(.*?[\s\*])(\w+)\s*(?:\[\d*\])?$", a, flags=re.S)
                assert m, a
                ty = _c_type_to_rust(m.group(1))
                if "[" in a:  # a parameter array is a pointer to its element
                    ty = "*mut " + ty
                params.append((m.group(2), ty))
        out[name] = (params, None if ret.strip() == "void" else _c_type_to_rust(ret))
    return out


def _norm_rust_type(t):
    t = re.sub(r"std::(ffi|os::raw)::", "", t.strip())
    return re.sub(r"\s+", " ", t)


def _rust_blocks():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 1."):md.index("## 2.")]
    text = "\n".join(re.findall(r"```rust\n(.*?)```", sec, flags=re.S))
    text = re.sub(r"//[^\n]*", "", text)
    structs, protos = {}, {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in [x.strip() for x in re.split(r",(?![^\[]*\])", body) if x.strip()]:
            m = re.match(r"pub (\w+)\s*:\s*(.+)$", f, flags=re.S)
            assert m, f
            fields.append((m.group(1), _norm_rust_type(m.group(2))))
        structs[name] = fields
    for name, args, ret in re.findall(r"pub fn (rt_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        params = []
        depth, cur = 0, ""
        for ch in args:  # split on top-level commas (the callback type has its own parentheses)
            depth += ch in "(<[" 
            depth -= ch in ")>]"
            if ch == "," and depth == 0:
                params.append(cur), (cur := "")
            else:
                cur += ch
        if cur.strip():
            params.append(cur)
        plist = []
        for a in params:
            pname, ty = a.split(":", 1)
            ty = _norm_rust_type(ty)
            plist.append((pname.strip(), "RtProgressFn" if ty.startswith("Option<extern") else ty))
        protos[name] = (plist, _norm_rust_type(ret) if ret else None)
    return structs, protos


def test_rust_binding_text_matches_the_headers():
    """INTEGRATION.md section 1 is the binding a maintainer of the reference pastes into src/gpu.rs; no rustc here checks it, so
    this does: every #[repr(C)] struct has the header's fields in the header's order with the corresponding types, and every
    `pub fn` the header's arguments (count, order, types) and return type.  Both headers, nothing missing on either side."""
    r_structs, r_protos = _rust_blocks()
    c_structs, c_protos = {}, {}
    for h in (PRODUCT_HEADER, DEBUG_HEADER):
        c_structs.update(_c_structs(h))
        c_protos.update(_c_prototypes(h))
    assert set(r_structs) == set(c_structs), set(r_structs) ^ set(c_structs)
    for name, fields in c_structs.items():
        assert r_structs[name] == fields, (name, [(a, b) for a, b in zip(r_structs[name], fields) if a != b])
    assert set(r_protos) == set(c_protos), set(r_protos) ^ set(c_protos)
    for name, (params, ret) in c_protos.items():
        rp, rr = r_protos[name]
        assert [t for _, t in rp] == [t for _, t in params], (name, rp, params)
        assert rr == ret, (name, rr, ret)
    # the first Rust block is the product header and nothing else
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    first = re.findall(r"```rust\n(.*?)```", md[md.index("## 1."):], flags=re.S)[0]
    assert sorted(re.findall(r"pub fn (rt_\w+)", first)) == _declared_functions(PRODUCT_HEADER)


def test_c_host_example_needs_only_the_product_headers(tmp_path):
    """examples/host_main.c compiles against rtow_host.h + rtow_mi355x.h in a directory that does not hold the debug header."""
    inc = tmp_path / "include"
    inc.mkdir()
    for h in ("rtow_host.h", PRODUCT_HEADER):
        (inc / h).write_text(open(os.path.join(ROOT, "include", h)).read())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", str(inc), "-c", os.path.join(ROOT, "examples", "host_main.c"),
                    "-o", str(tmp_path / "host_main.o")], check=True)


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
