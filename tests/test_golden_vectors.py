"""Oracle and GPU against tests/golden/bounce_vectors.json — the known answers of tests/golden/np_ref.py, a numpy
float32 restatement of one bounce written separately from oracle/oracle.cpp and from the HIP kernels (every
Material::scatter of material.rs / pbr.rs, Sphere::hit, the rectangles, Translate::hit, RotateY::hit,
ConstantMedium::hit, the Perlin turbulence on a fixed table, the ImageTex lookup, the counter RNG with its rejection
loop).  The reference ships no vectors of its own (parity
unpinned); this is the third, independent leg the three restatements are held to.

Exactness: hit, t, alive, scattered origin and direction are IEEE +,-,*,/,sqrt only and must match BIT FOR BIT;
so must the attenuation of every material that calls no libm function.  DisneyMetal (sin/cos of the rotation),
DisneyClearcoat (ln), the Perlin marble (sin) and the miss colour are compared to 2e-5 relative."""
import ctypes as C
import json
import os

import numpy as np
import pytest

VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bounce_vectors.json")))
LIBM_MATERIALS = {10, 12}


def f32s(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def _rays(rows):
    o = np.stack([f32s(r["o"]) for r in rows])
    d = np.stack([f32s(r["d"]) for r in rows])
    k = np.array([r["key"] for r in rows], dtype=np.uint32)
    return o, d, k


def _check(rows, got, exact_colour, production=False, t_rtol=0.0):
    want_hit = np.array([r["hit"] for r in rows])
    assert np.array_equal(np.minimum(got["hit"], 0) + (got["hit"] >= 0) * 0, np.where(want_hit < 0, -1, 0))  # one sphere: index 0 or a miss
    live = np.array([r["alive"] for r in rows], dtype=bool)
    hit = want_hit >= 0
    assert np.array_equal(got["alive"].astype(bool), live)
    want_t = np.array([r["t"] for r in rows], dtype=np.uint32)
    if t_rtol:  # a medium's scatter distance goes through ln(): last-ulp differences between libms move t and the origin
        assert np.allclose(got["t"][hit], want_t.view(np.float32)[hit], rtol=t_rtol, atol=0)
    else:
        assert np.array_equal(got["t"][hit].view(np.uint32), want_t[hit])
    for k, f in (("o", "so"), ("d", "sd")):
        want = np.array([r[f] for r in rows], dtype=np.uint32)
        if t_rtol and k == "o":
            assert np.allclose(got[k][live], want.view(np.float32)[live], rtol=t_rtol, atol=1e-6), k
        else:
            assert np.array_equal(got[k][live].view(np.uint32), want[live]), k
    att = np.array([r["att"] for r in rows], dtype=np.uint32)
    rad = np.array([r["rad"] for r in rows], dtype=np.uint32)
    # attenuation means something only when scatter() returned true (main.rs:48-54); the production path moreover
    # keeps no ray for a finished path and no radiance slot for a survivor
    sel_att = live
    sel_rad = ~live if production else np.ones(len(rows), dtype=bool)
    if exact_colour:
        assert np.array_equal(got["attenuation"][sel_att].view(np.uint32), att[sel_att])
        assert np.array_equal(got["radiance"][sel_rad & hit].view(np.uint32), rad[sel_rad & hit])
    else:
        assert np.allclose(got["attenuation"][sel_att], att.view(np.float32)[sel_att], rtol=2e-5, atol=1e-7)
        assert np.allclose(got["radiance"][sel_rad & hit], rad.view(np.float32)[sel_rad & hit], rtol=2e-5, atol=1e-7)
    assert np.allclose(got["radiance"][sel_rad & ~hit], rad.view(np.float32)[sel_rad & ~hit], rtol=2e-5, atol=1e-7)  # sky


def _sphere_scene(rt, mat_entry=None, tex="const", rot=None):
    s = rt.Scene.new()
    if tex == "perlin":
        t0 = s.perlin_tex(VEC["perlin"]["scale"])
    elif tex == "image":
        from ray_tracing_in_one_weekend_amd import images
        img = f32s(VEC["image"]["texels"]).reshape(VEC["image"]["h"], VEC["image"]["w"], 3)
        images.register_image("golden/image_tex", img)
        t0 = s.image_tex("golden/image_tex")
    else:
        t0 = s.constant_tex(VEC["tex"])
    t1 = s.constant_tex(VEC["tex1"])
    ty, prm = (mat_entry["type"], mat_entry["p"]) if mat_entry else ({"perlin": 1, "image": 0, "const": 1}[tex], (0, 0, 0, 0))
    m = s.material(ty, tex0=t0, tex1=t1, color=VEC["color"], p=prm)
    c = VEC["rotate_y"]["c"] if rot else VEC["sphere"]["c"]
    sp = s.sphere(c, VEC["sphere"]["r"], m, "golden")
    if rot:
        s.rotate_y(sp, rot)
    s.set_sky(rt._ffi.SKY_GRADIENT, None)
    s.set_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 16 / 9)
    return s.finish()


def _with_perlin_tables(rt, scene):
    """A copy of the flat scene whose Perlin set is the fixture's table (the scene API draws its own from the host RNG)."""
    fs = rt.RtFlatScene.from_buffer_copy(scene.flat)
    vec = np.ascontiguousarray(f32s(VEC["perlin"]["vec"]))
    perm = np.ascontiguousarray(np.array(VEC["perlin"]["perm"], dtype=np.uint16))
    fs.perlin_vec = vec.ctypes.data_as(C.POINTER(C.c_float))
    fs.perlin_perm = perm.ctypes.data_as(C.POINTER(C.c_uint16))
    return fs, (vec, perm)


def _rect_scene(rt):
    rc = VEC["rect_translate"]["rect"]
    s = rt.Scene.new()
    m = s.material(rt._ffi.MAT_DIFFUSE, tex0=s.constant_tex(VEC["tex"]))
    h = s.rect(rc["axis"], rc["min"], rc["max"], m)
    s.translate(h, rc["offset"])
    s.set_sky(rt._ffi.SKY_GRADIENT, None)
    s.set_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 16 / 9)
    return s.finish()


def _medium_scene(rt):
    s = rt.Scene.new()
    m = s.material(rt._ffi.MAT_DIFFUSE, tex0=s.constant_tex((0.5, 0.5, 0.5)))  # the boundary's own material is never shaded
    sp = s.sphere(VEC["sphere"]["c"], VEC["sphere"]["r"], m, "boundary")
    s.constant_medium(sp, VEC["medium"]["density"], s.constant_tex(VEC["tex"]))
    # a second, far-away object: alone in the world the medium would sit in a one-object BvhNode, which calls it twice per
    # visit (hitable.rs:188, 236-237) and so doubles its density — a quirk of the tree, not of hit()
    s.sphere((500.0, 500.0, 500.0), 0.25, m, "elsewhere")
    s.set_sky(rt._ffi.SKY_GRADIENT, None)
    s.set_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 16 / 9)
    return s.finish()


def _cases(rt):
    """(name, scene-or-flat, keepalive, rows, depth, exact colour?)"""
    for m in VEC["materials"]:
        yield f"material{m['type']}", _sphere_scene(rt, m), None, m["rays"], m["depth"], m["type"] not in LIBM_MATERIALS
    yield "rect_translate", _rect_scene(rt), None, VEC["rect_translate"]["rays"], VEC["rect_translate"]["depth"], True
    yield "medium", _medium_scene(rt), None, VEC["medium"]["rays"], VEC["medium"]["depth"], True
    sc = _sphere_scene(rt, rot=VEC["rotate_y"]["angle"])
    xf = sc.arrays()["xf_param"][:2].view(np.uint32)
    assert [int(xf[0]), int(xf[1])] == [VEC["rotate_y"]["sin"], VEC["rotate_y"]["cos"]], "sin/cos of the host mirror differ from numpy's"
    yield "rotate_y", sc, None, VEC["rotate_y"]["rays"], VEC["rotate_y"]["depth"], True
    sc = _sphere_scene(rt, tex="perlin")
    fs, keep = _with_perlin_tables(rt, sc)
    yield "perlin", fs, (sc, keep), VEC["perlin"]["rays"], VEC["perlin"]["depth"], False
    yield "image", _sphere_scene(rt, tex="image"), None, VEC["image"]["rays"], VEC["image"]["depth"], True


def test_oracle_matches_the_numpy_restatement(rt, orc):
    for name, scene, keep, rows, depth, exact in _cases(rt):
        o, d, k = _rays(rows)
        ptr = scene.flat_ptr if hasattr(scene, "flat_ptr") else C.pointer(scene)
        for accel in (orc.ACCEL_LIST, orc.ACCEL_BVH):
            got = orc.debug_bounce(ptr, o, d, k, depth=depth, accel=accel)
            try:
                _check(rows, got, exact, t_rtol=2e-6 if name == "medium" else 0.0)
            except AssertionError as e:
                raise AssertionError(f"{name}: {e}") from e


def test_oracle_image_lookup_and_rejection_loop(rt, orc):
    lib = orc.load()
    sc = _sphere_scene(rt, tex="image")
    tex = int(sc.arrays()["mat_tex0"][0])
    for case in VEC["image_lookup"]:
        uv, out = f32s(case["uv"]).copy(), np.zeros(3, np.float32)
        p = np.zeros(3, np.float32)
        f = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        lib.orc_texture_value(sc.flat_ptr, tex, f(uv), f(p), f(out))
        assert np.array_equal(out.view(np.uint32), np.array(case["rgb"], dtype=np.uint32)), case
    # random_in_unit_sphere on the counter generator: same accepted vector after the same number of draws
    for case in VEC["rejection"]:
        ctr = (case["depth"] + 1) * 256
        n = 0
        while True:
            u = [np.float32(lib.orc_ctr_draw(case["key"][0], case["key"][1], ctr + n + i) >> 8) * np.float32(1.0 / 16777216.0) for i in range(3)]
            n += 3
            v = [x * (np.float32(1.0) - np.float32(-1.0)) + np.float32(-1.0) for x in u]
            if (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2] < np.float32(1.0):
                break
        assert n == case["draws"] and [int(np.array([x], np.float32).view(np.uint32)[0]) for x in v] == case["v"]


@pytest.mark.gpu
def test_gpu_matches_the_numpy_restatement(rt, renderer):
    for name, scene, keep, rows, depth, exact in _cases(rt):
        o, d, k = _rays(rows)
        renderer.upload(scene)
        for flags, production in ((0, False), (rt._ffi.FLAG_BRUTE_FORCE, False), (rt._ffi.FLAG_PRODUCTION_KERNELS, True)):
            got = renderer.debug_bounce(o, d, k, depth=depth, flags=flags)
            try:
                _check(rows, got, exact, production, t_rtol=2e-6 if name == "medium" else 0.0)
            except AssertionError as e:
                raise AssertionError(f"{name} flags={flags}: {e}") from e
